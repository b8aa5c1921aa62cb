// CIN (Compressed Interaction Network, xDeepFM) - one layer, forward and backward,
// on the f32-input MFMA (v_mfma_f32_32x32x2_f32: exact fp32, 64 FLOP/clk/SIMD).
// Replaces the loop body of CIN.__call__ (recman/tf/core/layers.py:714-752):
//   Z[b,d,i*H+j] = X0[b,i,d]*Xk[b,j,d];  M = Z @ W + bias;  out = act(M) laid out [B,N,D]
//
// GEMM view: rows p = (b,d) (B*D of them), K = m*H, N filters.  The reference
// materialises Z (2.8 GB + 7.0 GB at the benchmark shape); here Z never exists:
// X0 and Xk of a 256-row block sit in LDS as [field][row] and every MFMA A-operand
// element is formed in a register by one multiply (1 v_mul per 64-cycle MFMA).
// The filter streams through LDS in 32-row chunks, double-buffered, one barrier
// per chunk.  MFMA-bound by construction: per 2 k's a wave issues 2*NT MFMAs
// (512 cycles of matrix pipe at NT=4) against 5 LDS reads and 2 multiplies.
//
// k' ordering: k' = i*He + j with He = H rounded up to even, so that the two k's
// of one MFMA (lane halves h=0/1) always share i.  The padded j = H row of Xk is
// zero in LDS and the matching filter rows are zero in the prepared filter.
#include <type_traits>

#include "rm_common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kRows = 256;  // (b,d) rows per forward block

__host__ __device__ inline int cin_He(int H) { return H + (H & 1); }
__host__ __device__ inline int cin_Kp(int m, int H) { return ((m * cin_He(H) + 31) / 32) * 32; }

// First layer (Xk IS X0, H = m): Z[i][j] = X0[i]*X0[j] is symmetric, so only the pairs j >= i are
// enumerated, against the folded filter W'[i][j] = W[i][j] + W[j][i] (j > i), W[i][i] (j = i):
// K' drops from m*He to about half (m = 26: 704 -> 384 padded) in the forward and the dW pass.
// Row i runs j from (i & ~1) to He-1 so that the two k' of an MFMA step still share i; the one
// extra element (j = i-1 for odd i) carries a zero filter row.
__host__ __device__ inline int cin_sym_len(int He, int i) { return He - (i & ~1); }
__host__ __device__ inline int cin_sym_start(int He, int i) {
  const int r = i >> 1;
  const int s = 2 * (r * He - r * (r - 1));
  return (i & 1) ? s + (He - 2 * r) : s;
}
__host__ __device__ inline int cin_Kp_sym(int m, int H) {
  const int He = cin_He(H);
  return ((cin_sym_start(He, m - 1) + cin_sym_len(He, m - 1) + 31) / 32) * 32;
}
// k' -> (i, j); i = m when k' lies past the last pair
__host__ __device__ inline void cin_sym_ij(int kp, int m, int He, int &i, int &j) {
  for (i = 0; i < m; ++i) {
    const int s0 = cin_sym_start(He, i);
    if (kp < s0 + cin_sym_len(He, i)) {
      j = (i & ~1) + (kp - s0);
      return;
    }
  }
  i = m;
  j = 0;
}
// the generic k' map of both orderings; ok = a real (i, j) pair of this ordering
__host__ __device__ inline void cin_kp_ij(int kp, int m, int H, int sym, int &i, int &j, bool &ok) {
  const int He = cin_He(H);
  if (sym) {
    cin_sym_ij(kp, m, He, i, j);
    ok = i < m && j >= i && j < H;
  } else {
    i = kp / He;
    j = kp - i * He;
    ok = i < m && j < H;
  }
}

__device__ __forceinline__ float act_apply(float v, int act) {
  if (act == RM_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == RM_ACT_LEAKY_RELU) return v > 0.f ? v : 0.2f * v;
  return v;
}
// act'(pre-activation) from the POST-activation value (same sign for relu/leaky)
__device__ __forceinline__ float act_grad_from_out(float o, int act) {
  if (act == RM_ACT_RELU) return o > 0.f ? 1.f : 0.f;
  if (act == RM_ACT_LEAKY_RELU) return o > 0.f ? 1.f : 0.2f;
  return 1.f;
}

// Wp[k'][c*NT + nt] = W[(i*H+j)*N + nt*32 + c]  (zero where padded): the B-operand
// layout - a lane reads its NT filter columns with one ds_read_b(32*NT).
__global__ void cin_prep_fwd_kernel(const float *__restrict__ W, int m, int H, int N, int NT, int sym,
                                    float *__restrict__ Wp) {
  const int Kp = sym ? cin_Kp_sym(m, H) : cin_Kp(m, H), Np = 32 * NT;
  const int total = Kp * Np;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
    const int kp = t / Np, r = t - kp * Np;
    const int c = r / NT, nt = r - c * NT;
    const int n = nt * 32 + c;
    int i, j;
    bool ok;
    cin_kp_ij(kp, m, H, sym, i, j, ok);
    float v = 0.f;
    if (ok && n < N) {
      v = W[(int64_t)(i * H + j) * N + n];
      if (sym && j > i) v += W[(int64_t)(j * H + i) * N + n];  // folded filter of the symmetric layer
    }
    Wp[t] = v;
  }
}

// MT = 32-row M-tiles per wave; the block always covers 256 rows with 8/MT waves.
// MT = 1 (8 waves, 2 per SIMD) lets one wave's s_waitcnt / barrier time be covered by its
// SIMD partner: rocprofv3 showed 26 % of wave-cycles parked in waits at 1 wave per SIMD.
#ifndef RM_CIN_EXP
#define RM_CIN_EXP 0  // ablation builds only (profiles/r01_p11): 1 = no per-chunk barrier, 2 = no output stores, 4 = non-temporal output stores (cin_fwd; none of them moves layer 1)
#endif
// Tile staging global -> LDS with ALL of a batch's loads issued before its first LDS store.  The plain
// `for (t = tid; ...) lds[...] = global[...]` loops compiled to load -> s_waitcnt -> store per iteration:
// 11 dependent round trips = 10.4 us of a 236 us cin_fwd block at layer 1, with every wave of the CU in the
// same phase (tools/probe/cin_clock.py).  src(t) -> float4 (zero for rows past B), dst(t) -> LDS address.
template <int NTHR, int BATCH, typename Src, typename Dst>
__device__ __forceinline__ void cin_stage(int total, int tid, Src src, Dst dst) {
  for (int base = tid; base < total; base += BATCH * NTHR) {
    float4 v[BATCH];
#pragma unroll
    for (int q = 0; q < BATCH; ++q) {
      const int t = base + q * NTHR;
      v[q] = src(t < total ? t : total - 1);
    }
#pragma unroll
    for (int q = 0; q < BATCH; ++q) {
      const int t = base + q * NTHR;
      if (t < total) *reinterpret_cast<float4 *>(dst(t)) = v[q];
    }
  }
}

// diagnostic build (-DRM_CIN_STAMP): thread 0 of every cin_fwd block records the shader clocks (s_memtime) and the
// 100 MHz ticks (s_memrealtime) its chunk loop took, and the block's total ticks; rm_debug_cin_stamps reads them
// (tools/probe/cin_clock.py).  Never defined in the product build.
#ifdef RM_CIN_STAMP
__device__ unsigned long long rm_cin_stamp_buf[4 * 8192];
#endif
template <int NT, int MT, int CPB = 1>
__global__ __launch_bounds__(512 / MT) void cin_fwd_kernel(
    const float *__restrict__ X0, const float *__restrict__ Xk, int64_t xk_bstride,
    const float *__restrict__ Wp, const float *__restrict__ bias, int act, int64_t B, int m, int H,
    int N, int D, float *__restrict__ out, float *__restrict__ pooled, int pool_stride,
    int pool_col0, int pool_from, int sym) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int Np = 32 * NT;
  constexpr int WCH = 32 * Np;  // floats per filter chunk
  const int He = cin_He(H);
  const int Kp = sym ? cin_Kp_sym(m, H) : cin_Kp(m, H);
  float *X0s = smem;                      // [(m+1)][256], row m is zero
  float *Xks = X0s + (m + 1) * kRows;     // [He][256], row H (if padded) is zero
  float *Ws = Xks + He * kRows;           // [2][32][Np]
  constexpr int NTHR = 512 / MT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, c = lane & 31;
  const int epb = kRows / D;
  const int64_t b0 = (int64_t)blockIdx.x * epb;
  const int D4 = D / 4;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#ifdef RM_CIN_STAMP
  const unsigned long long st_r0 = __builtin_amdgcn_s_memrealtime();
#endif

  cin_stage<NTHR, 4>(
      epb * m * D4, tid,
      [&](int t) {
        const int d4 = t % D4, i = (t / D4) % m, bl = t / (D4 * m);
        const int64_t b = b0 + bl;
        const float4 v = *reinterpret_cast<const float4 *>(X0 + ((b < B ? b : B - 1) * m + i) * D + 4 * d4);
        return b < B ? v : z4;
      },
      [&](int t) {
        const int d4 = t % D4, i = (t / D4) % m, bl = t / (D4 * m);
        return X0s + i * kRows + bl * D + 4 * d4;
      });
  if (tid < kRows) X0s[m * kRows + tid] = 0.f;
  cin_stage<NTHR, 8>(
      epb * H * D4, tid,
      [&](int t) {
        const int d4 = t % D4, j = (t / D4) % H, bl = t / (D4 * H);
        const int64_t b = b0 + bl;
        const float4 v =
            *reinterpret_cast<const float4 *>(Xk + (b < B ? b : B - 1) * xk_bstride + (int64_t)j * D + 4 * d4);
        return b < B ? v : z4;
      },
      [&](int t) {
        const int d4 = t % D4, j = (t / D4) % H, bl = t / (D4 * H);
        return Xks + j * kRows + bl * D + 4 * d4;
      });
  if (He > H && tid < kRows) Xks[H * kRows + tid] = 0.f;
  // CPB filter chunks of 32 k' are staged per barrier (CPB = 2 where the LDS allows: tools/probe/cin_clock.py
  // showed the chunk loop at 81 % MFMA-busy with a barrier per chunk and 94 % with none - the two waves of a
  // SIMD stall together at every barrier - so the barriers are halved: 2 buffers x CPB chunks)
  constexpr int PF = (NT * 256 + NTHR - 1) / NTHR;  // float4 of a filter chunk per thread
  {
    const int nch0 = Kp / 32;
#pragma unroll
    for (int cq = 0; cq < CPB; ++cq)
#pragma unroll
      for (int q = 0; q < PF; ++q)
        if (tid + q * NTHR < NT * 256)
          *reinterpret_cast<float4 *>(Ws + cq * WCH + (tid + q * NTHR) * 4) =
              *reinterpret_cast<const float4 *>(Wp + (int64_t)(cq < nch0 ? cq : nch0 - 1) * WCH + (tid + q * NTHR) * 4);
  }
  __syncthreads();

  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

#ifdef RM_CIN_STAMP
  const unsigned long long st_c1 = __builtin_amdgcn_s_memtime(), st_r1 = __builtin_amdgcn_s_memrealtime();
#endif
  const int prow = wave * 32 * MT + c;  // + mt*32
  const int nchunks = Kp / 32;
  const bool aligned = !sym && (He & 31) == 0;  // block-uniform
  int i_cur = 0, j_cur = 0;  // (i, even j) of the running k' pair: wave-uniform
  static_assert(PF <= 2, "the filter prefetch below is written out for at most 2 float4 per thread");
  const int pfi0 = tid < NT * 256 ? tid : NT * 256 - 1;                     // clamped: loads are
  const int pfi1 = tid + NTHR < NT * 256 ? tid + NTHR : NT * 256 - 1;       // always in bounds
  static_assert(CPB == 1 || CPB == 2, "filter prefetch registers are written out for 1 or 2 chunks per barrier");
  const int ngroups = (nchunks + CPB - 1) / CPB;
  for (int grp = 0; grp < ngroups; ++grp) {
    // The next group's filter chunks are prefetched UNCONDITIONALLY into named registers (chunk indices
    // clamped: the last group re-loads the last chunk).  As a conditional load into a float4 pf[] array the
    // prefetch lived in scratch: global_load -> s_waitcnt vmcnt(0) -> scratch_store, i.e. the
    // whole L2 latency exposed twice per chunk (rocprof: 26 % of wave-cycles waiting).
    const int na = (grp + 1) * CPB < nchunks ? (grp + 1) * CPB : nchunks - 1;
    const int nb = (grp + 1) * CPB + 1 < nchunks ? (grp + 1) * CPB + 1 : nchunks - 1;
    const float4 pf0 = *reinterpret_cast<const float4 *>(Wp + (int64_t)na * WCH + pfi0 * 4);
    float4 pf1 = pf0, pf2 = pf0, pf3 = pf0;
    if constexpr (PF > 1) pf1 = *reinterpret_cast<const float4 *>(Wp + (int64_t)na * WCH + pfi1 * 4);
    if constexpr (CPB > 1) {
      pf2 = *reinterpret_cast<const float4 *>(Wp + (int64_t)nb * WCH + pfi0 * 4);
      if constexpr (PF > 1) pf3 = *reinterpret_cast<const float4 *>(Wp + (int64_t)nb * WCH + pfi1 * 4);
    }
    __builtin_amdgcn_sched_barrier(0);  // keep the loads here, ahead of the MFMA steps
#pragma unroll
    for (int cq = 0; cq < CPB; ++cq) {
    const int ch = grp * CPB + cq;
    if (ch >= nchunks) break;  // block-uniform (odd chunk count)
    const float *Wb = Ws + (grp & 1) * (CPB * WCH) + cq * WCH;
    // Operands one k-step ahead.  Left to itself hipcc issued every step's filter read right before
    // its first MFMA (ds_read_b128 -> s_waitcnt lgkmcnt(0) -> 4 MFMAs): the whole LDS latency sat in
    // front of every group of four MFMAs and the kernel ran at 67 % with or without its per-chunk
    // barrier (ablation, profiles/r01_p11).  The sched_group_barriers pin the order
    // [next step's 3 reads] [this step's product] [this step's 4 MFMAs].
    auto read_step = [&](int sidx, float &x0v, float &xkv, float (&wv)[NT]) {
      const int irow = i_cur < m ? i_cur : m;
      x0v = X0s[irow * kRows + prow];
      xkv = Xks[(j_cur + h) * kRows + prow];
      const float *wp = Wb + (2 * sidx + h) * Np + c * NT;
      if constexpr (NT == 4) {
        const float4 t4 = *reinterpret_cast<const float4 *>(wp);
        wv[0] = t4.x; wv[1] = t4.y; wv[2] = t4.z; wv[3] = t4.w;
      } else if constexpr (NT == 2) {
        const float2 t2 = *reinterpret_cast<const float2 *>(wp);
        wv[0] = t2.x; wv[1] = t2.y;
      } else {
        wv[0] = wp[0];
      }
      j_cur += 2;  // (i, j) of the NEXT read
      if (j_cur >= He) { ++i_cur; j_cur = sym ? (i_cur & ~1) : 0; }
    };
    static_assert(MT == 1, "the pipelined step loop is written for one M-tile per wave");
    if (aligned) {
      // He % 32 == 0 (layer 1 of configs[2]: H = 64): a chunk of 32 k' = (i, j0 .. j0 + 31) has ONE i - x0 is
      // read once per chunk, the xk reads sit at compile-time offsets from one per-chunk base, and the
      // (i, j) bookkeeping is two scalar operations per chunk.  The generic loop below spends 3-5 vector
      // instructions per k-step on the two operand addresses; beside the partner wave's f32 MFMAs (which run
      // on the same vector ALU) that was ~8 % of the chunk time (tools/probe/cin_clock.py: 81 % MFMA-busy).
      const int cpi = He >> 5;  // chunks per i
      const int ic = ch / cpi, j0 = (ch - ic * cpi) << 5;
      const float x0v = X0s[ic * kRows + prow];
      const float *xkp = Xks + (j0 + h) * kRows + prow;
      auto read_fast = [&](int sidx, float &xkv, float (&wv)[NT]) {
        xkv = xkp[2 * sidx * kRows];
        const float *wp = Wb + (2 * sidx + h) * Np + c * NT;
        if constexpr (NT == 4) {
          const float4 t4 = *reinterpret_cast<const float4 *>(wp);
          wv[0] = t4.x; wv[1] = t4.y; wv[2] = t4.z; wv[3] = t4.w;
        } else if constexpr (NT == 2) {
          const float2 t2 = *reinterpret_cast<const float2 *>(wp);
          wv[0] = t2.x; wv[1] = t2.y;
        } else {
          wv[0] = wp[0];
        }
      };
      float xkc, xkn, wc[NT], wn[NT];
      read_fast(0, xkc, wc);
      __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        if (s + 1 < 16) read_fast(s + 1, xkn, wn);
        const float av = x0v * xkc;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, wc[nt], acc[0][nt], 0, 0, 0);
        if (s + 1 < 16) {
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);  // DS reads of step s + 1
          __builtin_amdgcn_sched_group_barrier(0x008, NT, 0); // MFMAs of step s
          xkc = xkn;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) wc[nt] = wn[nt];
        }
      }
    } else {
    float x0c, xkc, x0n, xkn, wc[NT], wn[NT];
    read_step(0, x0c, xkc, wc);
    __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);  // (the groups are filled in program order)
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      if (s + 1 < 16) read_step(s + 1, x0n, xkn, wn);
      const float av = x0c * xkc;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, wc[nt], acc[0][nt], 0, 0, 0);
      if (s + 1 < 16) {
        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);  // DS reads of step s + 1
        __builtin_amdgcn_sched_group_barrier(0x008, NT, 0); // MFMAs of step s
        x0c = x0n; xkc = xkn;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) wc[nt] = wn[nt];
      }
    }
    }  // generic (i, j) walk
    }  // chunks of the group
    {
      // stores are unconditional wherever the chunk divides evenly over the threads: a store
      // inside `if (tid + ... < NT*256)` let the compiler sink the prefetch LOAD into that branch
      float *Wn = Ws + ((grp + 1) & 1) * (CPB * WCH);
      constexpr bool kExact = (NT * 256) % NTHR == 0;
      if (kExact || tid < NT * 256) *reinterpret_cast<float4 *>(Wn + tid * 4) = pf0;
      if constexpr (PF > 1) {
        if (kExact || tid + NTHR < NT * 256) *reinterpret_cast<float4 *>(Wn + (tid + NTHR) * 4) = pf1;
      }
      if constexpr (CPB > 1) {
        if (kExact || tid < NT * 256) *reinterpret_cast<float4 *>(Wn + WCH + tid * 4) = pf2;
        if constexpr (PF > 1) {
          if (kExact || tid + NTHR < NT * 256) *reinterpret_cast<float4 *>(Wn + WCH + (tid + NTHR) * 4) = pf3;
        }
      }
    }
#if !(RM_CIN_EXP & 1)
    __syncthreads();
#endif
  }

#ifdef RM_CIN_STAMP
  const unsigned long long st_c2 = __builtin_amdgcn_s_memtime(), st_r2 = __builtin_amdgcn_s_memrealtime();
#endif
  // ---- epilogue: bias + activation, [B,N,D] store, pooled sums through LDS ----
  // pool_s [rows / 4][Np]: every (4-row group, filter) partial has its OWN cell (plain stores), and the
  // D / 4 partials of an (example, filter) are then added in row order: bit-reproducible.  (The first
  // version added them with a float atomicAdd into one cell per example - four adders in hardware order.)
  float *pool_s = Ws;  // 64 x Np floats <= 32 KB: fits the filter buffers
  const bool want_pool = pooled != nullptr;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = nt * 32 + c;
      const float bn = n < N ? bias[n] : 0.f;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int p = wave * 32 * MT + mt * 32 + 8 * g + 4 * h;  // first of 4 consecutive rows
        const int bl = p / D, d = p - bl * D;
        const int64_t b = b0 + bl;
        float4 v;
        v.x = act_apply(acc[mt][nt][4 * g + 0] + bn, act);
        v.y = act_apply(acc[mt][nt][4 * g + 1] + bn, act);
        v.z = act_apply(acc[mt][nt][4 * g + 2] + bn, act);
        v.w = act_apply(acc[mt][nt][4 * g + 3] + bn, act);
        if (b < B && n < N) {
#if RM_CIN_EXP & 4
          {
            typedef float f4v __attribute__((ext_vector_type(4)));
            f4v t4;
            t4.x = v.x; t4.y = v.y; t4.z = v.z; t4.w = v.w;
            __builtin_nontemporal_store(t4, reinterpret_cast<f4v *>(out + (b * N + n) * D + d));
          }
#elif !(RM_CIN_EXP & 2)
          *reinterpret_cast<float4 *>(out + (b * N + n) * D + d) = v;
#endif
          if (want_pool && n >= pool_from) pool_s[(p >> 2) * Np + n] = (v.x + v.y) + (v.z + v.w);
        }
      }
    }
  }
  if (want_pool) {
    __syncthreads();
    const int ncols = N - pool_from;
    for (int t = tid; t < epb * ncols; t += NTHR) {
      const int bl = t / ncols, cidx = t - bl * ncols;
      const int64_t b = b0 + bl;
      if (b < B) {
        const int q0 = bl * (D >> 2);
        float t4 = pool_s[q0 * Np + pool_from + cidx];
        for (int q = 1; q < (D >> 2); ++q) t4 += pool_s[(q0 + q) * Np + pool_from + cidx];
        pooled[b * pool_stride + pool_col0 + cidx] = t4;
      }
    }
  }
#ifdef RM_CIN_STAMP
  if (tid == 0 && blockIdx.x < 8192) {
    __builtin_amdgcn_s_waitcnt(0);
    rm_cin_stamp_buf[4 * blockIdx.x] = st_c2 - st_c1;
    rm_cin_stamp_buf[4 * blockIdx.x + 1] = st_r2 - st_r1;
    rm_cin_stamp_buf[4 * blockIdx.x + 2] = __builtin_amdgcn_s_memrealtime() - st_r0;
    rm_cin_stamp_buf[4 * blockIdx.x + 3] = st_r1 - st_r0;
  }
#endif
}
#ifdef RM_CIN_STAMP
}  // namespace
extern "C" int rm_debug_cin_stamps(unsigned long long *host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(rm_cin_stamp_buf), sizeof(unsigned long long) * n);
}
namespace {
#endif

// ===========================================================================
// Backward.
// ===========================================================================
// dM[p][n] = dOut[b,n,d] * act'(out[b,n,d]),  p = b*D + d, row-major [B*D][Np]
// (zero for n >= N).  dOut = the next layer's dXk for the "next hidden" half
// (n < pool_from) and g[b]*cin_w[col] for the direct-connect half (the gradient of
// reduce_sum + matmul, layers.py:754-758).  Also dbias[n] += sum_p dM[p][n].
// EPB examples per block iteration (EPB * D = 64 rows of dM): float4 loads of out / d_hidden (coalesced
// along d), a transposing trip through LDS, float4 stores of whole dM rows.  Four blocks fit a CU, so
// one block's load latency hides under the others' stores.  (The first version took ONE example per
// iteration with scalar accesses and two block barriers around 8 elements of work per thread: 465 us
// for 1.3 GB, 2.9 TB/s.)
__global__ __launch_bounds__(256) void cin_dm_kernel(
    const float *__restrict__ out, const float *__restrict__ d_hidden, int64_t dh_bstride,
    const float *__restrict__ g, const float *__restrict__ cw, int pool_from, int act, int64_t B,
    int N, int Np, int D, int EPB, float *__restrict__ dM, float *__restrict__ dbias_part) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int ld = Np + 4;
  float *tile = smem;                    // [EPB * D][ld]
  float *colsum = tile + EPB * D * ld;   // [Np]
  const int tid = threadIdx.x;
  const int D4 = D >> 2, ND4 = N * D4, Np4 = Np >> 2;
  float csum = 0.f;  // thread tid < Np: column tid
  // columns N..Np-1 of the tile are zero for good (dM's padding columns)
  for (int t = tid; t < EPB * D * (Np - N); t += 256) {
    const int r = t / (Np - N), n = N + (t - r * (Np - N));
    tile[r * ld + n] = 0.f;
  }
  const int64_t ngroups = (B + EPB - 1) / EPB;
  for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int64_t b0 = grp * EPB;
    __syncthreads();  // the previous group's tile has been stored
    for (int t = tid; t < EPB * ND4; t += 256) {
      const int e = t / ND4, r = t - e * ND4, n = r / D4, d4 = r - n * D4;
      const int64_t b = b0 + e;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (b < B) {
        const float4 o = *reinterpret_cast<const float4 *>(out + (b * N + n) * D + 4 * d4);
        float4 up;
        if (n >= pool_from) {
          const float u = g[b] * cw[n - pool_from];
          up = make_float4(u, u, u, u);
        } else {
          up = *reinterpret_cast<const float4 *>(d_hidden + b * dh_bstride + (int64_t)n * D + 4 * d4);
        }
        v = make_float4(up.x * act_grad_from_out(o.x, act), up.y * act_grad_from_out(o.y, act),
                        up.z * act_grad_from_out(o.z, act), up.w * act_grad_from_out(o.w, act));
      }
      float *tp = tile + (e * D + 4 * d4) * ld + n;
      tp[0] = v.x; tp[ld] = v.y; tp[2 * ld] = v.z; tp[3 * ld] = v.w;
    }
    __syncthreads();
    for (int t = tid; t < EPB * D * Np4; t += 256) {
      const int r = t / Np4, n4 = t - r * Np4;  // r = e * D + d
      const int64_t b = b0 + r / D;
      if (b < B)
        *reinterpret_cast<float4 *>(dM + (b0 * D + r) * Np + 4 * n4) =
            *reinterpret_cast<const float4 *>(tile + r * ld + 4 * n4);
    }
    if (tid < Np) {
      float s0 = 0.f, s1 = 0.f;
      for (int r = 0; r < EPB * D; r += 2) {
        s0 += tile[r * ld + tid];
        s1 += tile[(r + 1) * ld + tid];
      }
      csum += s0 + s1;
    }
  }
  // per-block partial column sums; cin_dbias_reduce_kernel adds them in block order
  // (deterministic - a float atomicAdd here made cin_bias gradients differ in the last bits
  // between two identical steps)
  (void)colsum;
  if (tid < Np) dbias_part[(int64_t)blockIdx.x * Np + tid] = csum;
}

// one block per column n: 256 threads take every 256th partial, then a fixed-order tree in LDS
// (a single thread walking all 2048 partials took 596 us)
__global__ __launch_bounds__(256) void cin_dbias_reduce_kernel(const float *__restrict__ part, int nblk,
                                                               int N, int Np, float *__restrict__ dbias) {
  __shared__ float sm[256];
  const int n = blockIdx.x, tid = threadIdx.x;
  float s = 0.f;
  for (int i = tid; i < nblk; i += 256) s += part[(int64_t)i * Np + n];
  sm[tid] = s;
  __syncthreads();
#pragma unroll
  for (int w = 128; w > 0; w >>= 1) {
    if (tid < w) sm[tid] += sm[tid + w];
    __syncthreads();
  }
  if (tid == 0) dbias[n] = sm[0];
}

// dX-kernel k' ordering: k' = i*Hp + j with Hp = H rounded up to 32, so that every
// 32-row k'-tile has ONE i and a compile-time j per accumulator register.
__host__ __device__ inline int cin_Hp(int H) { return H <= 32 ? 32 : (H <= 64 ? 64 : 128); }

// Wq[k'][n] = W[(i*H+j)*N + n] (zero where padded): the A-operand source of the dX kernel.
__global__ void cin_prep_bwd_kernel(const float *__restrict__ W, int m, int H, int N, int Np,
                                    float *__restrict__ Wq) {
  const int Hp = cin_Hp(H);
  const int total = m * Hp * Np;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
    const int kp = t / Np, n = t - kp * Np;
    const int i = kp / Hp, j = kp - i * Hp;
    Wq[t] = (j < H && n < N) ? W[(int64_t)(i * H + j) * N + n] : 0.f;
  }
}

// rows per dX block: 256 (8 waves, 2 per SIMD: waits of one wave are covered by its SIMD
// partner) when the LDS images fit, else 128 (4 waves)

// dZ^T tile [32 k'][32 rows] = W[k'][:] . dM[row][:]^T on the MFMA (A = filter chunk from
// LDS, B = the lane's own dM row kept in registers), contracted on the fly IN REGISTERS:
//   dX0[i][row] = sum_j dZ * Xk[j][row]   (scalar per lane per i; the two lane halves
//                                          are added with one cross-lane move)
//   dXk[j][row] += dZ * X0[i][row]        (JB*16 registers per lane, j static per register)
// No LDS atomics: an earlier version contracted with ds_add_f32 and ran 8x slower.
template <int NT, int JB, int ROWS>
__global__ __launch_bounds__(ROWS * 2) void cin_dx_kernel(
    const float *__restrict__ X0, const float *__restrict__ Xk, int64_t xk_bstride, int xk_is_x0,
    const float *__restrict__ Wq, const float *__restrict__ dM, int64_t B, int m, int H, int D,
    float *__restrict__ dX0, int accumulate_dx0, float *__restrict__ dXk, int64_t dxk_bstride) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int Np = 32 * NT;
  constexpr int LDW = Np + 4;
  constexpr int Hp = 32 * JB;
  constexpr int kRowsX = ROWS;
  constexpr int NTHR = ROWS * 2;
  float *X0s = smem;                         // [m][ROWS]
  float *Xks = X0s + m * kRowsX;             // [Hp][128] (rows >= H zero)
  float *dX0s = Xks + Hp * kRowsX;           // [m][128]
  float *Wt = dX0s + m * kRowsX;             // [2][32][LDW]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, c = lane & 31;
  const int epb = kRowsX / D;
  const int64_t b0 = (int64_t)blockIdx.x * epb;
  const int D4 = D / 4;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);

  cin_stage<NTHR, 4>(
      epb * m * D4, tid,
      [&](int t) {
        const int d4 = t % D4, i = (t / D4) % m, bl = t / (D4 * m);
        const int64_t b = b0 + bl;
        const float4 v = *reinterpret_cast<const float4 *>(X0 + ((b < B ? b : B - 1) * m + i) * D + 4 * d4);
        return b < B ? v : z4;
      },
      [&](int t) {
        const int d4 = t % D4, i = (t / D4) % m, bl = t / (D4 * m);
        return X0s + i * kRowsX + bl * D + 4 * d4;
      });
  cin_stage<NTHR, 8>(
      epb * Hp * D4, tid,
      [&](int t) {
        const int d4 = t % D4, j = (t / D4) % Hp, bl = t / (D4 * Hp);
        const int64_t b = b0 + bl;
        const float4 v = *reinterpret_cast<const float4 *>(Xk + (b < B ? b : B - 1) * xk_bstride +
                                                           (int64_t)(j < H ? j : H - 1) * D + 4 * d4);
        return (b < B && j < H) ? v : z4;
      },
      [&](int t) {
        const int d4 = t % D4, j = (t / D4) % Hp, bl = t / (D4 * Hp);
        return Xks + j * kRowsX + bl * D + 4 * d4;
      });
  // the lane's own dM row, in MFMA B-operand order: step s=4u+q <-> n = 8u + 4h + q
  const int prow = wave * 32 + c;
  const int64_t pg = b0 * D + prow;  // global row
  float dm[Np / 2];
#pragma unroll
  for (int u = 0; u < Np / 8; ++u) {
    const float4 v = pg < B * D ? *reinterpret_cast<const float4 *>(dM + pg * Np + 8 * u + 4 * h) : z4;
    dm[4 * u + 0] = v.x; dm[4 * u + 1] = v.y; dm[4 * u + 2] = v.z; dm[4 * u + 3] = v.w;
  }
  // filter chunk 0
  constexpr int CF4 = 32 * Np / 4;                 // float4 per filter chunk
  constexpr int F4 = (CF4 + NTHR - 1) / NTHR;      // per thread
#pragma unroll
  for (int q = 0; q < F4; ++q) {
    const int f = tid + q * NTHR, row = f / (Np / 4), c4 = f - row * (Np / 4);
    if (f < CF4)
      *reinterpret_cast<float4 *>(Wt + row * LDW + 4 * c4) =
          *reinterpret_cast<const float4 *>(Wq + (int64_t)row * Np + 4 * c4);
  }
  __syncthreads();

  // Xk values of this lane's (row, j) pairs and the dXk accumulators: register r of
  // j-block jb <-> j = jb*32 + (r&3) + 8*(r>>2) + 4h
  float xkr[JB][16], dxk[JB][16];
#pragma unroll
  for (int jb = 0; jb < JB; ++jb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      xkr[jb][r] = Xks[(jb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * kRowsX + prow];
      dxk[jb][r] = 0.f;
    }

  const int ntiles = m * JB;
  int kt = 0;
  for (int i = 0; i < m; ++i) {
    const float x0v = X0s[i * kRowsX + prow];
    float dx0 = 0.f;
#pragma unroll
    for (int jb = 0; jb < JB; ++jb, ++kt) {
      // next filter tile -> NAMED registers, unconditionally (the last tile re-loads itself,
      // threads past the tile re-load its last float4): the conditional float4 pf[F4] array
      // of the first version lived in scratch behind an s_waitcnt vmcnt(0) per load
      static_assert(F4 <= 4, "the filter prefetch is written out for at most 4 float4 per thread");
      const int ktn = kt + 1 < ntiles ? kt + 1 : kt;
      const float *wsrc = Wq + (int64_t)ktn * 32 * Np;
#define RM_PF(q)                                                                         \
  float4 pf##q = z4;                                                                     \
  if constexpr (q < F4) {                                                                \
    const int f_ = (tid + q * NTHR) < CF4 ? (tid + q * NTHR) : CF4 - 1;                  \
    pf##q = *reinterpret_cast<const float4 *>(wsrc + (int64_t)(f_ / (Np / 4)) * Np + 4 * (f_ % (Np / 4))); \
  }
      RM_PF(0) RM_PF(1) RM_PF(2) RM_PF(3)
#undef RM_PF
      __builtin_amdgcn_sched_barrier(0);
      const float *Wb = Wt + (kt & 1) * 32 * LDW;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int u = 0; u < Np / 8; ++u) {
        const float4 a4 = *reinterpret_cast<const float4 *>(Wb + c * LDW + 8 * u + 4 * h);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, dm[4 * u + 0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, dm[4 * u + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, dm[4 * u + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, dm[4 * u + 3], acc, 0, 0, 0);
      }
      // acc[r] = dZ[row = prow][(i, j = jb*32 + (r&3) + 8*(r>>2) + 4h)]
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        dx0 += acc[r] * xkr[jb][r];
        dxk[jb][r] += acc[r] * x0v;
      }
      {
        float *Wn = Wt + ((kt + 1) & 1) * 32 * LDW;
        constexpr bool kExact = CF4 % NTHR == 0;  // unconditional stores keep the loads from sinking
#define RM_ST(q)                                                                         \
  if constexpr (q < F4) {                                                                \
    const int f_ = tid + q * NTHR;                                                       \
    if (kExact || f_ < CF4)                                                              \
      *reinterpret_cast<float4 *>(Wn + (f_ / (Np / 4)) * LDW + 4 * (f_ % (Np / 4))) = pf##q; \
  }
        RM_ST(0) RM_ST(1) RM_ST(2) RM_ST(3)
#undef RM_ST
      }
      __syncthreads();
    }
    dx0 += __shfl_xor(dx0, 32, 64);  // the two lane halves hold disjoint j sets
    if (h == 0) dX0s[i * kRowsX + prow] = dx0;
  }
  // dXk -> LDS (reusing the Xk image: every (j,row) is owned by exactly one lane)
#pragma unroll
  for (int jb = 0; jb < JB; ++jb)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      Xks[(jb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * kRowsX + prow] = dxk[jb][r];
  __syncthreads();

  for (int t = tid; t < epb * m * D4; t += NTHR) {
    const int d4 = t % D4, i = (t / D4) % m, bl = t / (D4 * m);
    const int64_t b = b0 + bl;
    if (b >= B) continue;
    float4 v = *reinterpret_cast<const float4 *>(dX0s + i * kRowsX + bl * D + 4 * d4);
    if (xk_is_x0) {
      const float4 w = *reinterpret_cast<const float4 *>(Xks + i * kRowsX + bl * D + 4 * d4);
      v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
    }
    float4 *dst = reinterpret_cast<float4 *>(dX0 + (b * m + i) * D + 4 * d4);
    if (accumulate_dx0) {
      const float4 o = *dst;
      v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
    }
    *dst = v;
  }
  if (!xk_is_x0 && dXk != nullptr) {
    for (int t = tid; t < epb * H * D4; t += NTHR) {
      const int d4 = t % D4, j = (t / D4) % H, bl = t / (D4 * H);
      const int64_t b = b0 + bl;
      if (b >= B) continue;
      *reinterpret_cast<float4 *>(dXk + b * dxk_bstride + (int64_t)j * D + 4 * d4) =
          *reinterpret_cast<const float4 *>(Xks + j * kRowsX + bl * D + 4 * d4);
    }
  }
}

// First layer (Xk IS X0): dX0[a] = sum_b dZs[a][b] x0[b] with the SYMMETRIC dZs[a][b] = dM . (W[a][b] +
// W[b][a]) - only the pairs b >= a are formed (the k' ordering of the forward / dW pass, K' = 384
// instead of 26 x 32 = 832 at m = 26), pair (i, j) then feeds dX0[i] += dZs x0[j] and dX0[j] += dZs x0[i]
// (i = j: both land on the same field, 2 dZs x0[i], as d(x^2) wants).
// Wq[k'][n] = folded filter row of pair k' (zero rows where the ordering pads).
__global__ void cin_prep_bwd_sym_kernel(const float *__restrict__ W, int m, int H, int N, int Np,
                                        float *__restrict__ Wq) {
  const int total = cin_Kp_sym(m, H) * Np;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
    const int kp = t / Np, n = t - kp * Np;
    int i, j;
    bool ok;
    cin_kp_ij(kp, m, H, 1, i, j, ok);
    float v = 0.f;
    if (ok && n < N) {
      v = W[(int64_t)(i * H + j) * N + n];
      if (j > i) v += W[(int64_t)(j * H + i) * N + n];
    }
    Wq[t] = v;
  }
}

// Same MFMA scheme as cin_dx_kernel (A = filter tile from LDS, B = the lane's dM row in registers), but a
// tile's 32 k' are arbitrary (i, j) pairs, so the contraction with x0 cannot use static registers: each
// lane walks the 16 pairs it holds for ITS row (the two lane halves hold disjoint halves of the tile) -
// the i side accumulates in a register while i stays the same (pairs run j-fastest), the j side is a
// read-modify-write of the half's own accumulator image dXa[h][field][row] (no atomics: a (half, row) has
// one owner).  The two images are added at the end.
template <int NT>
__global__ __launch_bounds__(512) void cin_dx_sym_kernel(
    const float *__restrict__ X0, const float *__restrict__ Wq, const float *__restrict__ dM, int64_t B,
    int m, int D, float *__restrict__ dX0, int accumulate_dx0) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int Np = 32 * NT;
  constexpr int LDW = Np + 4;
  constexpr int ROWS = 256, NTHR = 512;
  const int He = cin_He(m), Kp = cin_Kp_sym(m, m), ntiles = Kp / 32;
  float *X0s = smem;                         // [m][ROWS]
  float *dXa = X0s + m * ROWS;               // [2][m][ROWS]
  float *Wt = dXa + 2 * m * ROWS;            // [2][32][LDW]
  int *ijs = reinterpret_cast<int *>(Wt + 2 * 32 * LDW);  // [Kp]: i | j << 8
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, c = lane & 31;
  const int epb = ROWS / D;
  const int64_t b0 = (int64_t)blockIdx.x * epb;
  const int D4 = D / 4;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);

  for (int t = tid; t < epb * m * D4; t += NTHR) {
    const int d4 = t % D4, i = (t / D4) % m, bl = t / (D4 * m);
    const int64_t b = b0 + bl;
    const float4 v = b < B ? *reinterpret_cast<const float4 *>(X0 + (b * m + i) * D + 4 * d4) : z4;
    *reinterpret_cast<float4 *>(X0s + i * ROWS + bl * D + 4 * d4) = v;
  }
  for (int t = tid; t < 2 * m * ROWS; t += NTHR) dXa[t] = 0.f;
  for (int kp = tid; kp < Kp; kp += NTHR) {
    int i, j;
    cin_sym_ij(kp, m, He, i, j);
    // padding pairs (past the end, or the j = i - 1 / j = H slots of the ordering) carry a zero filter
    // row: any in-range field will do for them
    i = i < m ? i : m - 1;
    j = j < m ? j : m - 1;
    ijs[kp] = i | (j << 8);
  }
  const int prow = wave * 32 + c;
  const int64_t pg = b0 * D + prow;  // global row
  float dm[Np / 2];
#pragma unroll
  for (int u = 0; u < Np / 8; ++u) {
    const float4 v = pg < B * D ? *reinterpret_cast<const float4 *>(dM + pg * Np + 8 * u + 4 * h) : z4;
    dm[4 * u + 0] = v.x; dm[4 * u + 1] = v.y; dm[4 * u + 2] = v.z; dm[4 * u + 3] = v.w;
  }
  constexpr int CF4 = 32 * Np / 4;                 // float4 per filter tile
  constexpr int F4 = (CF4 + NTHR - 1) / NTHR;      // per thread
  static_assert(F4 <= 2, "the filter prefetch is written out for at most 2 float4 per thread");
#pragma unroll
  for (int q = 0; q < F4; ++q) {
    const int f = tid + q * NTHR, row = f / (Np / 4), c4 = f - row * (Np / 4);
    if (f < CF4)
      *reinterpret_cast<float4 *>(Wt + row * LDW + 4 * c4) =
          *reinterpret_cast<const float4 *>(Wq + (int64_t)row * Np + 4 * c4);
  }
  __syncthreads();

  float *dXh = dXa + h * m * ROWS;
  // acc[r] = dZs[row = prow][k' = kt*32 + (r&3) + 8*(r>>2) + 4h]: the lane holds, for ITS row, 16 of the
  // tile's 32 pairs (the other lane half holds the other 16) - contracted straight from the registers.
  auto contract = [&](const f32x16 &acc, int kt) {
    int cur_i = -1;
    float acc_i = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ij = ijs[kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h];
      const int i = ij & 255, j = ij >> 8;
      const float dz = acc[r];
      const float xi = X0s[i * ROWS + prow], xj = X0s[j * ROWS + prow];
      if (i != cur_i) {  // (uniform within the half)
        if (cur_i >= 0) dXh[cur_i * ROWS + prow] += acc_i;
        cur_i = i;
        acc_i = 0.f;
      }
      acc_i += dz * xj;
      dXh[j * ROWS + prow] += dz * xi;
    }
    dXh[cur_i * ROWS + prow] += acc_i;
  };
  // (Contracting tile kt-1 in the same straight-line block as tile kt's MFMAs - a software pipeline -
  // measured the same: 2.56 vs 2.54 ms for the layer's backward.)
  for (int kt = 0; kt < ntiles; ++kt) {
    // next filter tile -> named registers, unconditionally (see cin_dx_kernel)
    const int ktn = kt + 1 < ntiles ? kt + 1 : kt;
    const float *wsrc = Wq + (int64_t)ktn * 32 * Np;
    const int f0 = tid < CF4 ? tid : CF4 - 1, f1 = tid + NTHR < CF4 ? tid + NTHR : CF4 - 1;
    const float4 pf0 = *reinterpret_cast<const float4 *>(wsrc + (int64_t)(f0 / (Np / 4)) * Np + 4 * (f0 % (Np / 4)));
    float4 pf1 = pf0;
    if constexpr (F4 > 1)
      pf1 = *reinterpret_cast<const float4 *>(wsrc + (int64_t)(f1 / (Np / 4)) * Np + 4 * (f1 % (Np / 4)));
    __builtin_amdgcn_sched_barrier(0);
    const float *Wb = Wt + (kt & 1) * 32 * LDW;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int u = 0; u < Np / 8; ++u) {
      const float4 a4 = *reinterpret_cast<const float4 *>(Wb + c * LDW + 8 * u + 4 * h);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, dm[4 * u + 0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, dm[4 * u + 1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, dm[4 * u + 2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, dm[4 * u + 3], acc, 0, 0, 0);
    }
    contract(acc, kt);
    {
      float *Wn = Wt + ((kt + 1) & 1) * 32 * LDW;
      constexpr bool kExact = CF4 % NTHR == 0;
      if (kExact || tid < CF4) *reinterpret_cast<float4 *>(Wn + (tid / (Np / 4)) * LDW + 4 * (tid % (Np / 4))) = pf0;
      if constexpr (F4 > 1) {
        const int f_ = tid + NTHR;
        if (kExact || f_ < CF4) *reinterpret_cast<float4 *>(Wn + (f_ / (Np / 4)) * LDW + 4 * (f_ % (Np / 4))) = pf1;
      }
    }
    __syncthreads();
  }

  for (int t = tid; t < epb * m * D4; t += NTHR) {
    const int d4 = t % D4, i = (t / D4) % m, bl = t / (D4 * m);
    const int64_t b = b0 + bl;
    if (b >= B) continue;
    const float4 u0 = *reinterpret_cast<const float4 *>(dXa + i * ROWS + bl * D + 4 * d4);
    const float4 u1 = *reinterpret_cast<const float4 *>(dXa + (m + i) * ROWS + bl * D + 4 * d4);
    float4 v = make_float4(u0.x + u1.x, u0.y + u1.y, u0.z + u1.z, u0.w + u1.w);
    float4 *dst = reinterpret_cast<float4 *>(dX0 + (b * m + i) * D + 4 * d4);
    if (accumulate_dx0) {
      const float4 o = *dst;
      v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
    }
    *dst = v;
  }
}

size_t cin_dx_sym_smem(int m, int NT) {
  const int Np = 32 * NT;
  return (size_t)(3 * m * 256 + 2 * 32 * (Np + 4)) * sizeof(float) + (size_t)cin_Kp_sym(m, m) * sizeof(int);
}

size_t cin_dx_smem(int m, int H, int NT, int rows) {
  const int Np = 32 * NT;
  return (size_t)((2 * m + cin_Hp(H)) * rows + 2 * 32 * (Np + 4)) * sizeof(float);
}

// dW partial: part[s][k'][n] = sum over the split's rows p of Z[p][k'] * dM[p][n].
// MFMA with the rows as the reduction dimension: A = Z^T (formed in registers from the
// [row][field] LDS images), B = dM rows.  A wave owns 4 k'-tiles x NT n-tiles.
constexpr int kDmBlocks = 256 * 8;  // grid cap of the dM pass (= rows of its dbias partials)
constexpr int kRC = 64;   // rows per staged chunk
constexpr int kKT = 2;    // k'-tiles per wave
constexpr int kDWW = 8;   // waves per dW block (2 per SIMD)
constexpr int kMaxGroups = 16;

// Work plan of the dW pass: group g (16 k'-tiles) is split over S[g] row ranges of cps[g]
// chunks each; S[g] is proportional to the group's per-SIMD tile load so that every block
// takes the same time (a partially filled last group gets fewer, longer blocks).
struct DwPlan {
  int ngroups, Smax;
  int S[kMaxGroups], cps[kMaxGroups], blk0[kMaxGroups + 1];
};

DwPlan cin_dw_plan(int Kp, int64_t chunks_total, int target_blocks) {
  DwPlan p;
  const int ntiles = Kp / 32;
  p.ngroups = (ntiles + kDWW * kKT - 1) / (kDWW * kKT);
  int u[kMaxGroups], usum = 0;
  for (int g = 0; g < p.ngroups; ++g) {
    int t = ntiles - g * kDWW * kKT;
    if (t > kDWW * kKT) t = kDWW * kKT;
    int per_simd[4] = {0, 0, 0, 0};
    for (int i = 0; i < t; ++i) per_simd[(i % kDWW) % 4]++;
    // cost of one row chunk for this group, in hundredths of an MFMA tile-unit per SIMD; never below
    // 1.56: a chunk's staging (global loads, transposed LDS writes, two barriers) keeps lighter blocks
    // staging-bound (weighting a 1-unit group as 1 made its blocks the critical path, 4.5 -> 6.1 ms; as 2,
    // its blocks finished 22 % early: per-block timestamps at H = 64, tools/probe/cin_dw_stamps.py -
    // 6.61 us per chunk against 17.0 us for a 4-unit group)
    u[g] = 156;
    for (int q = 0; q < 4; ++q) u[g] = per_simd[q] * 100 > u[g] ? per_simd[q] * 100 : u[g];
    usum += u[g];
  }
  p.Smax = 0;
  p.blk0[0] = 0;
  for (int g = 0; g < p.ngroups; ++g) {
    int64_t S = (int64_t)target_blocks * u[g] / usum;
    if (S < 1) S = 1;
    if (S > chunks_total) S = chunks_total;
    if (S < 1) S = 1;
    p.S[g] = (int)S;
    p.cps[g] = (int)((chunks_total + S - 1) / S);
    p.blk0[g + 1] = p.blk0[g] + p.S[g];
    if (p.S[g] > p.Smax) p.Smax = p.S[g];
  }
  return p;
}
#ifdef RM_CIN_STAMP
__device__ unsigned long long rm_cin_dw_stamp_buf[4 * 1024];
#endif
template <int NT>
__global__ __launch_bounds__(512) void cin_dw_kernel(
    const float *__restrict__ X0, const float *__restrict__ Xk, int64_t xk_bstride,
    const float *__restrict__ dM, int64_t B, int m, int H, int D, DwPlan plan,
    float *__restrict__ part, int sym) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int Np = 32 * NT;
  constexpr int NTHR = 64 * kDWW;
  const int He = cin_He(H);
  const int Kp = sym ? cin_Kp_sym(m, H) : cin_Kp(m, H);
  const int ld0 = (m + 1) | 1, ldk = He | 1;
  float *X0T = smem;                 // [64][ld0]  (column m is zero)
  float *XkT = X0T + kRC * ld0;      // [64][ldk]
  float *dMs = XkT + kRC * ldk;      // [64][Np], n = nt*32+cc stored at cc*NT+nt
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, c = lane & 31;
#ifdef RM_CIN_STAMP
  const unsigned long long dwt0 = __builtin_amdgcn_s_memrealtime();
#endif
  // split-major block order: the groups of one row range run side by side and share the
  // staged rows through L2 / Infinity Cache (group-major order re-read them from HBM)
  int group = 0, split = 0;
  {
    int bleft = blockIdx.x;
    for (split = 0; split < plan.Smax; ++split) {
      int cnt = 0;
      for (int g = 0; g < plan.ngroups; ++g) cnt += plan.S[g] > split;
      if (bleft < cnt) break;
      bleft -= cnt;
    }
    for (group = 0; group < plan.ngroups; ++group) {
      if (plan.S[group] > split) {
        if (bleft == 0) break;
        --bleft;
      }
    }
  }
  const int chunks_per_split = plan.cps[group];
  const int64_t rows_total = B * D;
  const int64_t chunk0 = (int64_t)split * chunks_per_split;

  // tile -> wave round-robin (tile = group*16 + q*8 + wave): a partially filled last group
  // then loads every SIMD equally, and a wave skips the MFMAs of a tile slot beyond K'
  // (its SIMD partner gets the matrix pipe) instead of multiplying zeros.
  int iq[kKT], jq[kKT];
  bool tile_ok[kKT];
#pragma unroll
  for (int q = 0; q < kKT; ++q) {
    const int tile = group * kDWW * kKT + q * kDWW + wave;
    tile_ok[q] = tile * 32 < Kp;
    const int kp = tile * 32 + c;
    int i, j;
    bool ok;
    cin_kp_ij(kp, m, H, sym, i, j, ok);
    ok = ok && kp < Kp;
    iq[q] = ok ? i : m;
    jq[q] = ok ? j : 0;
  }
  f32x16 acc[kKT][NT];
#pragma unroll
  for (int q = 0; q < kKT; ++q)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[q][nt][r] = 0.f;

  const int D4 = D / 4;
  const int n0f4 = kRC * m / 4;   // float4 of X0 per chunk
  const int nkf4 = kRC * H / 4;
  constexpr int DMF4 = kRC * Np / 4 / NTHR;  // float4 of dM per thread per chunk
  constexpr int XF4 = 2;                      // X0 / Xk float4 per thread per chunk (m, H <= 64)
  float4 p0[XF4], pk[XF4], pd[DMF4];

  auto prefetch = [&](int64_t r0) {
#pragma unroll
    for (int q = 0; q < XF4; ++q) {
      const int t = tid + q * NTHR;
      p0[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      pk[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (t < n0f4) {
        const int d4 = t % D4, i = (t / D4) % m, bl = t / (D4 * m);
        const int64_t pgl = r0 + bl * D + 4 * d4;
        if (pgl < rows_total) p0[q] = *reinterpret_cast<const float4 *>(X0 + ((pgl / D) * m + i) * D + 4 * d4);
      }
      if (t < nkf4) {
        const int d4 = t % D4, j = (t / D4) % H, bl = t / (D4 * H);
        const int64_t pgl = r0 + bl * D + 4 * d4;
        if (pgl < rows_total)
          pk[q] = *reinterpret_cast<const float4 *>(Xk + (pgl / D) * xk_bstride + (int64_t)j * D + 4 * d4);
      }
    }
#pragma unroll
    for (int q = 0; q < DMF4; ++q) {
      const int f = tid + q * NTHR, pl = f / (Np / 4), c4 = f - pl * (Np / 4);
      const int64_t pgl = r0 + pl;
      pd[q] = pgl < rows_total ? *reinterpret_cast<const float4 *>(dM + pgl * Np + 4 * c4)
                               : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int q = 0; q < XF4; ++q) {
      const int t = tid + q * NTHR;
      if (t < n0f4) {
        const int d4 = t % D4, i = (t / D4) % m, bl = t / (D4 * m);
        const int pl = bl * D + 4 * d4;
        X0T[(pl + 0) * ld0 + i] = p0[q].x; X0T[(pl + 1) * ld0 + i] = p0[q].y;
        X0T[(pl + 2) * ld0 + i] = p0[q].z; X0T[(pl + 3) * ld0 + i] = p0[q].w;
      }
      if (t < nkf4) {
        const int d4 = t % D4, j = (t / D4) % H, bl = t / (D4 * H);
        const int pl = bl * D + 4 * d4;
        XkT[(pl + 0) * ldk + j] = pk[q].x; XkT[(pl + 1) * ldk + j] = pk[q].y;
        XkT[(pl + 2) * ldk + j] = pk[q].z; XkT[(pl + 3) * ldk + j] = pk[q].w;
      }
    }
#pragma unroll
    for (int q = 0; q < DMF4; ++q) {
      const int f = tid + q * NTHR, pl = f / (Np / 4), c4 = f - pl * (Np / 4);
      const int n = 4 * c4, nt = n >> 5, cc = n & 31;
      float *dst = dMs + pl * Np + cc * NT + nt;
      dst[0] = pd[q].x; dst[NT] = pd[q].y; dst[2 * NT] = pd[q].z; dst[3 * NT] = pd[q].w;
    }
  };

  if (tid < kRC) X0T[tid * ld0 + m] = 0.f;
  if (chunk0 * kRC < rows_total) prefetch(chunk0 * kRC);
  for (int ci = 0; ci < chunks_per_split; ++ci) {
    const int64_t r0 = (chunk0 + ci) * kRC;
    if (r0 >= rows_total) break;
    __syncthreads();  // previous chunk fully consumed
    commit();
    __syncthreads();
    // the next chunk's global loads fly during this chunk's 32 MFMA steps
    if (ci + 1 < chunks_per_split && r0 + kRC < rows_total) prefetch(r0 + kRC);
    // branch-free hot loop when both tile slots are real; the checked variant only runs in
    // the partially filled last group
    auto steps = [&](auto checked) {
      // unroll 2, not 4: at 4 the kernel spilled 4 VGPRs (row pointers), and their scratch reloads
      // in the prefetch put an s_waitcnt vmcnt(0) between the chunk's global loads
#pragma unroll 2
      for (int t = 0; t < kRC / 2; ++t) {
        const int pl = 2 * t + h;
        float bv[NT];
        const float *bp = dMs + pl * Np + c * NT;
        if constexpr (NT == 4) {
          const float4 t4 = *reinterpret_cast<const float4 *>(bp);
          bv[0] = t4.x; bv[1] = t4.y; bv[2] = t4.z; bv[3] = t4.w;
        } else if constexpr (NT == 2) {
          const float2 t2 = *reinterpret_cast<const float2 *>(bp);
          bv[0] = t2.x; bv[1] = t2.y;
        } else {
          bv[0] = bp[0];
        }
#pragma unroll
        for (int q = 0; q < kKT; ++q) {
          if constexpr (decltype(checked)::value) {
            if (!tile_ok[q]) continue;  // wave-uniform
          }
          const float a = X0T[pl * ld0 + iq[q]] * XkT[pl * ldk + jq[q]];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[q][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv[nt], acc[q][nt], 0, 0, 0);
        }
      }
    };
    if (tile_ok[0] && tile_ok[1]) steps(std::false_type{});
    else steps(std::true_type{});
  }
  // ---- partial slab: part[split][k'][n] ----
#pragma unroll
  for (int q = 0; q < kKT; ++q) {
    const int tile = group * kDWW * kKT + q * kDWW + wave;
    if (tile * 32 >= Kp) continue;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kp = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        part[((int64_t)split * Kp + kp) * Np + nt * 32 + c] = acc[q][nt][r];
      }
  }
#ifdef RM_CIN_STAMP
  if (tid == 0 && blockIdx.x < 1024) {
    rm_cin_dw_stamp_buf[4 * blockIdx.x] = dwt0;
    rm_cin_dw_stamp_buf[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
    rm_cin_dw_stamp_buf[4 * blockIdx.x + 2] = group;
    rm_cin_dw_stamp_buf[4 * blockIdx.x + 3] = chunks_per_split;
  }
#endif
}
#ifdef RM_CIN_STAMP
}  // namespace
extern "C" int rm_debug_cin_dw_stamps(unsigned long long *host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(rm_cin_dw_stamp_buf), sizeof(unsigned long long) * n);
}
namespace {
#endif

size_t cin_dw_smem(int m, int H, int NT) {
  return (size_t)(kRC * (((m + 1) | 1) + (cin_He(H) | 1) + 32 * NT)) * sizeof(float);
}

// dW[(i*H+j)][n] = sum_s part[s][k'(i,j)][n]; in the symmetric ordering (i,j) and (j,i) read the
// same k' (dLoss/dW[i][j] = dLoss/dW[j][i] = dLoss/dW'[min][max])
// 16 waves per block: 64 consecutive outputs, wave w adds up every 16th slab (a thread's chain of
// loads is S / 16 long instead of S), fixed-order tree in LDS -> deterministic
constexpr int kDwrG = 16;
__global__ __launch_bounds__(64 * kDwrG) void cin_dw_reduce_kernel(const float *__restrict__ part, DwPlan plan,
                                                                   int m, int H, int N, int Np, int sym,
                                                                   float *__restrict__ dW) {
  __shared__ float sm[kDwrG][64];
  const int He = cin_He(H), Kp = sym ? cin_Kp_sym(m, H) : cin_Kp(m, H);
  const int total = m * H * N;
  const int o = threadIdx.x & 63, grp = threadIdx.x >> 6;
  for (int t0 = blockIdx.x * 64; t0 < total; t0 += gridDim.x * 64) {
    const int t = t0 + o;
    float acc = 0.f;
    if (t < total) {
      const int k = t / N, n = t - k * N;
      const int i = k / H, j = k - i * H;
      int kp;
      if (sym) {
        const int lo = i < j ? i : j, hi = i < j ? j : i;
        kp = cin_sym_start(He, lo) + (hi - (lo & ~1));
      } else {
        kp = i * He + j;
      }
      const int S = plan.S[(kp / 32) / (kDWW * kKT)];
#pragma unroll 4
      for (int s = grp; s < S; s += kDwrG) acc += part[((int64_t)s * Kp + kp) * Np + n];
    }
    sm[grp][o] = acc;
    __syncthreads();
    if (grp == 0 && t < total) {
      float v[kDwrG];
#pragma unroll
      for (int q = 0; q < kDwrG; ++q) v[q] = sm[q][o];
#pragma unroll
      for (int st = 1; st < kDwrG; st *= 2)
#pragma unroll
        for (int q = 0; q < kDwrG; q += 2 * st) v[q] += v[q + st];
      dW[t] = v[0];
    }
    __syncthreads();
  }
}

size_t cin_fwd_smem(int m, int H, int NT, int cpb = 1) {
  return (size_t)((m + 1) * kRows + cin_He(H) * kRows + 2 * cpb * 32 * 32 * NT) * sizeof(float);
}

}  // namespace

extern "C" int64_t rm_cin_filter_workspace(int m, int H, int N) {
  const int NT = N <= 32 ? 1 : (N <= 64 ? 2 : 4);
  return (int64_t)cin_Kp(m, H) * 32 * NT;
}

// csrc/cin6.hip: the dX part on the bf16 matrix pipe with split fp32 operands
int64_t rm_internal_cin_dx6_floats(int m, int H, int N, int D);
bool rm_internal_cin_dx6(const float *X0, const float *Xk, int64_t xk_bstride, int xk_is_x0, const float *W,
                         const float *dM, int64_t B, int m, int H, int N, int D, float *dX0, int accumulate_dx0,
                         float *dXk, int64_t dxk_bstride, float *ws6, hipStream_t st);
int64_t rm_internal_cin_dw6_floats(int64_t B, int m, int H, int N, int D);
bool rm_internal_cin_dw6(const float *X0, const float *Xk, int64_t xk_bstride, const float *dM, int64_t B, int m, int H,
                         int N, int D, float *dW, float *ws6, hipStream_t st);

static int cin_check(const char *fn, int64_t B, int m, int H, int N, int D) {
  RM_REQUIRE(B >= 0 && m > 0 && H > 0 && N > 0 && D > 0, "%s: bad sizes", fn);
  RM_REQUIRE(N <= 128, "%s: N=%d unsupported (<= 128 filters per layer)", fn, N);
  RM_REQUIRE(m <= 64, "%s: m=%d unsupported (<= 64 fields)", fn, m);
  RM_REQUIRE(H <= 128, "%s: H=%d unsupported (<= 128 hidden maps)", fn, H);
  RM_REQUIRE(D % 4 == 0 && kRows % D == 0, "%s: D=%d unsupported (must divide 256, multiple of 4)", fn, D);
  return RM_OK;
}

extern "C" int rm_cin_layer_fwd(const float *X0, const float *Xk, int64_t xk_bstride,
                                const float *W, const float *bias, int act, int64_t B, int m, int H,
                                int N, int D, float *out, float *pooled, int pool_stride,
                                int pool_col0, int pool_from, float *filter_ws, rm_stream_t stream) {
  int rc = cin_check("rm_cin_layer_fwd", B, m, H, N, D);
  if (rc != RM_OK) return rc;
  if (B == 0) return RM_OK;
  RM_REQUIRE(X0 && Xk && W && bias && out && filter_ws, "rm_cin_layer_fwd: NULL argument");
  RM_REQUIRE(rm_aligned16(X0) && rm_aligned16(Xk) && rm_aligned16(out) && rm_aligned16(filter_ws) &&
                 xk_bstride % 4 == 0,
             "rm_cin_layer_fwd: 16-byte alignment required");
  RM_REQUIRE(act >= RM_ACT_IDENTITY && act <= RM_ACT_LEAKY_RELU, "rm_cin_layer_fwd: bad activation id");
  RM_REQUIRE(!pooled || (pool_from >= 0 && pool_from <= N && pool_stride >= pool_col0 + N - pool_from),
             "rm_cin_layer_fwd: bad pooled layout");
  const int NT = N <= 32 ? 1 : (N <= 64 ? 2 : 4);
  const int cpb = cin_fwd_smem(m, H, NT, 2) <= 160 * 1024 ? 2 : 1;  // two filter chunks per barrier where they fit
  const size_t smem = cin_fwd_smem(m, H, NT, cpb);
  RM_REQUIRE(smem <= 160 * 1024, "rm_cin_layer_fwd: m=%d H=%d needs %zu B of LDS (> 160 KiB)", m, H, smem);
  hipStream_t st = (hipStream_t)stream;
  // first layer (Xk is X0 itself): symmetric k' ordering with a folded filter, about half the K'
  const int sym = (Xk == X0 && H == m && xk_bstride == (int64_t)m * D) ? 1 : 0;
  hipLaunchKernelGGL(cin_prep_fwd_kernel, dim3(256), dim3(256), 0, st, W, m, H, N, NT, sym, filter_ws);
  const int epb = kRows / D;
  dim3 grid((unsigned)((B + epb - 1) / epb));
#define RM_CIN_FWD(NT_, CPB_)                                                                         \
  {                                                                                                   \
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(cin_fwd_kernel<NT_, 1, CPB_>),           \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);                 \
    hipLaunchKernelGGL((cin_fwd_kernel<NT_, 1, CPB_>), grid, dim3(512), smem, st, X0, Xk, xk_bstride, \
                       filter_ws, bias, act, B, m, H, N, D, out, pooled, pool_stride, pool_col0,      \
                       pool_from, sym);                                                               \
  }
  if (cpb == 2) {
    if (NT == 1) RM_CIN_FWD(1, 2) else if (NT == 2) RM_CIN_FWD(2, 2) else RM_CIN_FWD(4, 2)
  } else {
    if (NT == 1) RM_CIN_FWD(1, 1) else if (NT == 2) RM_CIN_FWD(2, 1) else RM_CIN_FWD(4, 1)
  }
#undef RM_CIN_FWD
  RM_CHECK_LAUNCH("rm_cin_layer_fwd");
  return RM_OK;
}

// floats of the dW partial slabs: the larger of the two k' orderings (the symmetric one has fewer
// tiles but may be split into more slabs)
static int64_t cin_part_floats(int64_t B, int m, int H, int Np, int D) {
  const int64_t chunks_total = (B * D + kRC - 1) / kRC;
  const int64_t ct = chunks_total > 0 ? chunks_total : 1;
  const int Kp = cin_Kp(m, H);
  int64_t need = (int64_t)cin_dw_plan(Kp, ct, 256).Smax * Kp * Np;
  if (H == m) {
    const int Ks = cin_Kp_sym(m, H);
    const int64_t ns = (int64_t)cin_dw_plan(Ks, ct, 256).Smax * Ks * Np;
    need = ns > need ? ns : need;
  }
  return need;
}

extern "C" int64_t rm_cin_bwd_workspace(int64_t B, int m, int H, int N, int D) {
  const int NT = N <= 32 ? 1 : (N <= 64 ? 2 : 4);
  const int Np = 32 * NT;
  return (int64_t)m * cin_Hp(H) * Np + B * D * Np + cin_part_floats(B, m, H, Np, D) + (int64_t)kDmBlocks * Np +
         rm_internal_cin_dx6_floats(m, H, N, D) + rm_internal_cin_dw6_floats(B, m, H, N, D);
}
extern "C" int rm_cin_layer_bwd(const float *X0, const float *Xk, int64_t xk_bstride, int xk_is_x0,
                                const float *W, int act, const float *out, const float *d_hidden,
                                int64_t dh_bstride, const float *g, const float *cin_w_direct,
                                int pool_from, int64_t B, int m, int H, int N, int D, float *dX0,
                                int accumulate_dx0, float *dXk, int64_t dxk_bstride, float *dW,
                                float *dbias, float *workspace, int64_t workspace_floats,
                                rm_stream_t stream) {
  int rc = cin_check("rm_cin_layer_bwd", B, m, H, N, D);
  if (rc != RM_OK) return rc;
  if (B == 0) return RM_OK;
  RM_REQUIRE(kRC % D == 0, "rm_cin_layer_bwd: D=%d unsupported (must divide 64)", D);
  RM_REQUIRE(X0 && Xk && W && out && g && dX0 && dW && dbias && workspace,
             "rm_cin_layer_bwd: NULL argument");
  RM_REQUIRE(pool_from >= 0 && pool_from <= N, "rm_cin_layer_bwd: bad pool_from");
  RM_REQUIRE(pool_from == 0 || d_hidden, "rm_cin_layer_bwd: d_hidden needed when pool_from > 0");
  RM_REQUIRE(pool_from == N || cin_w_direct, "rm_cin_layer_bwd: cin_w_direct needed");
  RM_REQUIRE(xk_is_x0 || dXk, "rm_cin_layer_bwd: dXk needed unless Xk is X0");
  RM_REQUIRE(!xk_is_x0 || H == m, "rm_cin_layer_bwd: xk_is_x0 needs H == m");
  RM_REQUIRE(workspace_floats >= rm_cin_bwd_workspace(B, m, H, N, D),
             "rm_cin_layer_bwd: workspace too small");
  RM_REQUIRE(rm_aligned16(X0) && rm_aligned16(Xk) && rm_aligned16(dX0) && rm_aligned16(workspace) &&
                 (!dXk || rm_aligned16(dXk)) && xk_bstride % 4 == 0 && dxk_bstride % 4 == 0,
             "rm_cin_layer_bwd: 16-byte alignment required");
  const int NT = N <= 32 ? 1 : (N <= 64 ? 2 : 4);
  const int Np = 32 * NT, Kp = cin_Kp(m, H);
  float *Wq = workspace;
  float *dM = Wq + (int64_t)m * cin_Hp(H) * Np;
  float *part = dM + B * D * Np;
  hipStream_t st = (hipStream_t)stream;
  float *dbias_part = part + cin_part_floats(B, m, H, Np, D);  // [kDmBlocks][Np]

#ifndef RM_CIN_DX_SYM
#define RM_CIN_DX_SYM 1
#endif
  // first layer: the symmetric-pair dX kernel (about half the MFMA work), when its LDS images fit
  const bool split = (accumulate_dx0 & 2) != 0;  // bit 1: dX / dW on the bf16 pipe where csrc/cin6.hip covers the layer
  // bit 2: the FIRST layer's dX on the split-operand kernel too (full k' range, H padded to 32) instead of the
  // symmetric-pair f32 kernel with half the work
  const bool first6 = split && (accumulate_dx0 & 4) != 0 && rm_internal_cin_dx6_floats(m, H, N, D) > 0;
  const bool dx_sym = !first6 && RM_CIN_DX_SYM && xk_is_x0 && H == m && 256 % D == 0 && m <= 255 &&
                      cin_dx_sym_smem(m, NT) <= 160 * 1024;
  if (dx_sym) hipLaunchKernelGGL(cin_prep_bwd_sym_kernel, dim3(256), dim3(256), 0, st, W, m, H, N, Np, Wq);
  else hipLaunchKernelGGL(cin_prep_bwd_kernel, dim3(256), dim3(256), 0, st, W, m, H, N, Np, Wq);
  {
    int epb = 64 / D;  // EPB * D = 64 dM rows per iteration (D <= 64; D = 4, 8: capped at 8 examples)
    epb = epb < 1 ? 1 : (epb > 8 ? 8 : epb);
    const size_t smem = (size_t)(epb * D * (Np + 4) + Np) * sizeof(float);
    const int nblk = rm_grid_cap((B + epb - 1) / epb, kDmBlocks);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(cin_dm_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(cin_dm_kernel, dim3(nblk), dim3(256), smem, st, out, d_hidden, dh_bstride, g,
                       cin_w_direct, pool_from, act, B, N, Np, D, epb, dM, dbias_part);
    hipLaunchKernelGGL(cin_dbias_reduce_kernel, dim3(N), dim3(256), 0, st, dbias_part, nblk, N, Np, dbias);
  }
  if (dx_sym) {
    const size_t smem = cin_dx_sym_smem(m, NT);
    dim3 grid((unsigned)((B + 256 / D - 1) / (256 / D)));
#define RM_CIN_DXS(NT_)                                                                          \
  {                                                                                              \
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(cin_dx_sym_kernel<NT_>),            \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);            \
    hipLaunchKernelGGL((cin_dx_sym_kernel<NT_>), grid, dim3(512), smem, st, X0, Wq, dM, B, m, D,  \
                       dX0, accumulate_dx0 & 1);                                                 \
  }
    if (NT == 1) RM_CIN_DXS(1) else if (NT == 2) RM_CIN_DXS(2) else RM_CIN_DXS(4)
#undef RM_CIN_DXS
  } else if (split && (!xk_is_x0 || first6) &&
             rm_internal_cin_dx6(X0, Xk, xk_bstride, xk_is_x0, W, dM, B, m, H, N, D, dX0, accumulate_dx0 & 1, dXk,
                                 dxk_bstride, dbias_part + (((int64_t)kDmBlocks * Np + 3) / 4 * 4), st)) {
    // (bit 1 of accumulate_dx0: the dX part on the bf16 matrix pipe with split operands, csrc/cin6.hip)
  } else {
    accumulate_dx0 &= 1;
    const int rows = (kRC * 2 % D == 0 && 256 % D == 0 && cin_dx_smem(m, H, NT, 256) <= 160 * 1024) ? 256 : 128;
    const size_t smem = cin_dx_smem(m, H, NT, rows);
    RM_REQUIRE(smem <= 160 * 1024, "rm_cin_layer_bwd: m=%d H=%d needs %zu B of LDS (> 160 KiB)", m, H, smem);
    const int epb = rows / D;
    dim3 grid((unsigned)((B + epb - 1) / epb));
#define RM_CIN_DX(NT_, JB_, R_)                                                               \
  {                                                                                           \
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(cin_dx_kernel<NT_, JB_, R_>),    \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);         \
    hipLaunchKernelGGL((cin_dx_kernel<NT_, JB_, R_>), grid, dim3(R_ * 2), smem, st, X0, Xk,   \
                       xk_bstride, xk_is_x0, Wq, dM, B, m, H, D, dX0, accumulate_dx0, dXk,    \
                       dxk_bstride);                                                          \
  }
#define RM_CIN_DX_JB(NT_)                                                                      \
  {                                                                                            \
    if (JB == 1) { if (rows == 256) RM_CIN_DX(NT_, 1, 256) else RM_CIN_DX(NT_, 1, 128) }        \
    else if (JB == 2) { if (rows == 256) RM_CIN_DX(NT_, 2, 256) else RM_CIN_DX(NT_, 2, 128) }   \
    else RM_CIN_DX(NT_, 4, 128)                                                                \
  }
    const int JB = cin_Hp(H) / 32;  // 1, 2 or 4
    if (NT == 1) RM_CIN_DX_JB(1) else if (NT == 2) RM_CIN_DX_JB(2) else RM_CIN_DX_JB(4)
#undef RM_CIN_DX_JB
#undef RM_CIN_DX
  }
  float *ws_dw6 = dbias_part + (((int64_t)kDmBlocks * Np + 3) / 4 * 4) + (rm_internal_cin_dx6_floats(m, H, N, D) + 3) / 4 * 4;
  if (split && !xk_is_x0 && rm_internal_cin_dw6(X0, Xk, xk_bstride, dM, B, m, H, N, D, dW, ws_dw6, st)) {
    // (the dW pass on the bf16 matrix pipe with split operands, csrc/cin6.hip)
  } else {
    const size_t smem = cin_dw_smem(m, H, NT);
    const int64_t chunks_total = (B * D + kRC - 1) / kRC;
    const int sym = xk_is_x0 ? 1 : 0;  // first layer: symmetric k' ordering (half the tiles)
    const DwPlan plan = cin_dw_plan(sym ? cin_Kp_sym(m, H) : Kp, chunks_total, 256);
    dim3 grid(plan.blk0[plan.ngroups]);
#define RM_CIN_DW(NT_)                                                                        \
  {                                                                                           \
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(cin_dw_kernel<NT_>),             \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);         \
    hipLaunchKernelGGL((cin_dw_kernel<NT_>), grid, dim3(512), smem, st, X0, Xk, xk_bstride,   \
                       dM, B, m, H, D, plan, part, sym);                                      \
  }
    if (NT == 1) RM_CIN_DW(1) else if (NT == 2) RM_CIN_DW(2) else RM_CIN_DW(4)
#undef RM_CIN_DW
    hipLaunchKernelGGL(cin_dw_reduce_kernel, dim3(rm_grid_cap(((int64_t)m * H * N + 63) / 64, 4096)),
                       dim3(64 * kDwrG), 0, st, part, plan, m, H, N, Np, sym, dW);
  }
  RM_CHECK_LAUNCH("rm_cin_layer_bwd");
  return RM_OK;
}
