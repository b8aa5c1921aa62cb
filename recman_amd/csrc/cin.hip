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
#include "rm_common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kRows = 256;  // (b,d) rows per forward block

__host__ __device__ inline int cin_He(int H) { return H + (H & 1); }
__host__ __device__ inline int cin_Kp(int m, int H) { return ((m * cin_He(H) + 31) / 32) * 32; }

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
__global__ void cin_prep_fwd_kernel(const float *__restrict__ W, int m, int H, int N, int NT,
                                    float *__restrict__ Wp) {
  const int He = cin_He(H), Kp = cin_Kp(m, H), Np = 32 * NT;
  const int total = Kp * Np;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
    const int kp = t / Np, r = t - kp * Np;
    const int c = r / NT, nt = r - c * NT;
    const int n = nt * 32 + c;
    const int i = kp / He, j = kp - i * He;
    Wp[t] = (i < m && j < H && n < N) ? W[(int64_t)(i * H + j) * N + n] : 0.f;
  }
}

template <int NT>
__global__ __launch_bounds__(256, 1) void cin_fwd_kernel(
    const float *__restrict__ X0, const float *__restrict__ Xk, int64_t xk_bstride,
    const float *__restrict__ Wp, const float *__restrict__ bias, int act, int64_t B, int m, int H,
    int N, int D, float *__restrict__ out, float *__restrict__ pooled, int pool_stride,
    int pool_col0, int pool_from) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int Np = 32 * NT;
  constexpr int WCH = 32 * Np;  // floats per filter chunk
  const int He = cin_He(H);
  const int Kp = cin_Kp(m, H);
  float *X0s = smem;                      // [(m+1)][256], row m is zero
  float *Xks = X0s + (m + 1) * kRows;     // [He][256], row H (if padded) is zero
  float *Ws = Xks + He * kRows;           // [2][32][Np]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, c = lane & 31;
  const int epb = kRows / D;
  const int64_t b0 = (int64_t)blockIdx.x * epb;
  const int D4 = D / 4;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);

  for (int t = tid; t < epb * m * D4; t += 256) {
    const int d4 = t % D4, i = (t / D4) % m, bl = t / (D4 * m);
    const int64_t b = b0 + bl;
    const float4 v = b < B ? *reinterpret_cast<const float4 *>(X0 + (b * m + i) * D + 4 * d4) : z4;
    *reinterpret_cast<float4 *>(X0s + i * kRows + bl * D + 4 * d4) = v;
  }
  X0s[m * kRows + tid] = 0.f;
  for (int t = tid; t < epb * H * D4; t += 256) {
    const int d4 = t % D4, j = (t / D4) % H, bl = t / (D4 * H);
    const int64_t b = b0 + bl;
    const float4 v =
        b < B ? *reinterpret_cast<const float4 *>(Xk + b * xk_bstride + (int64_t)j * D + 4 * d4) : z4;
    *reinterpret_cast<float4 *>(Xks + j * kRows + bl * D + 4 * d4) = v;
  }
  if (He > H) Xks[H * kRows + tid] = 0.f;
#pragma unroll
  for (int q = 0; q < NT; ++q)
    *reinterpret_cast<float4 *>(Ws + (tid + q * 256) * 4) =
        *reinterpret_cast<const float4 *>(Wp + (tid + q * 256) * 4);
  __syncthreads();

  f32x16 acc[2][NT];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  const int prow = wave * 64 + c;  // + mt*32
  const int nchunks = Kp / 32;
  int i_cur = 0, j_cur = 0;  // (i, even j) of the running k' pair: wave-uniform
  for (int ch = 0; ch < nchunks; ++ch) {
    float4 pf[NT];
    if (ch + 1 < nchunks) {
#pragma unroll
      for (int q = 0; q < NT; ++q)
        pf[q] = *reinterpret_cast<const float4 *>(Wp + (int64_t)(ch + 1) * WCH + (tid + q * 256) * 4);
    }
    const float *Wb = Ws + (ch & 1) * WCH;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int irow = i_cur < m ? i_cur : m;
      const float *x0p = X0s + irow * kRows + prow;
      const float *xkp = Xks + (j_cur + h) * kRows + prow;
      const float a0 = x0p[0] * xkp[0];
      const float a1 = x0p[32] * xkp[32];
      float bv[NT];
      const float *wp = Wb + (2 * s + h) * Np + c * NT;
      if constexpr (NT == 4) {
        const float4 t4 = *reinterpret_cast<const float4 *>(wp);
        bv[0] = t4.x; bv[1] = t4.y; bv[2] = t4.z; bv[3] = t4.w;
      } else if constexpr (NT == 2) {
        const float2 t2 = *reinterpret_cast<const float2 *>(wp);
        bv[0] = t2.x; bv[1] = t2.y;
      } else {
        bv[0] = wp[0];
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv[nt], acc[0][nt], 0, 0, 0);
        acc[1][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv[nt], acc[1][nt], 0, 0, 0);
      }
      j_cur += 2;
      if (j_cur >= He) { j_cur = 0; ++i_cur; }
    }
    if (ch + 1 < nchunks) {
      float *Wn = Ws + ((ch + 1) & 1) * WCH;
#pragma unroll
      for (int q = 0; q < NT; ++q) *reinterpret_cast<float4 *>(Wn + (tid + q * 256) * 4) = pf[q];
    }
    __syncthreads();
  }

  // ---- epilogue: bias + activation, [B,N,D] store, pooled sums through LDS ----
  float *pool_s = Ws;  // [epb][Np] (<= 32 KB, fits the two filter buffers when epb <= 64)
  const bool want_pool = pooled != nullptr;
  if (want_pool) {
    for (int t = tid; t < epb * Np; t += 256) pool_s[t] = 0.f;
    __syncthreads();
  }
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = nt * 32 + c;
      const float bn = n < N ? bias[n] : 0.f;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int p = wave * 64 + mt * 32 + 8 * g + 4 * h;  // first of 4 consecutive rows
        const int bl = p / D, d = p - bl * D;
        const int64_t b = b0 + bl;
        float4 v;
        v.x = act_apply(acc[mt][nt][4 * g + 0] + bn, act);
        v.y = act_apply(acc[mt][nt][4 * g + 1] + bn, act);
        v.z = act_apply(acc[mt][nt][4 * g + 2] + bn, act);
        v.w = act_apply(acc[mt][nt][4 * g + 3] + bn, act);
        if (b < B && n < N) {
          *reinterpret_cast<float4 *>(out + (b * N + n) * D + d) = v;
          if (want_pool && n >= pool_from) atomicAdd(pool_s + bl * Np + n, v.x + v.y + v.z + v.w);
        }
      }
    }
  }
  if (want_pool) {
    __syncthreads();
    const int ncols = N - pool_from;
    for (int t = tid; t < epb * ncols; t += 256) {
      const int bl = t / ncols, cidx = t - bl * ncols;
      const int64_t b = b0 + bl;
      if (b < B) pooled[b * pool_stride + pool_col0 + cidx] = pool_s[bl * Np + pool_from + cidx];
    }
  }
}

size_t cin_fwd_smem(int m, int H, int NT) {
  return (size_t)((m + 1) * kRows + cin_He(H) * kRows + 2 * 32 * 32 * NT) * sizeof(float);
}

}  // namespace

extern "C" int64_t rm_cin_filter_workspace(int m, int H, int N) {
  const int NT = N <= 32 ? 1 : (N <= 64 ? 2 : 4);
  return (int64_t)cin_Kp(m, H) * 32 * NT;
}

static int cin_check(const char *fn, int64_t B, int m, int H, int N, int D) {
  RM_REQUIRE(B >= 0 && m > 0 && H > 0 && N > 0 && D > 0, "%s: bad sizes", fn);
  RM_REQUIRE(N <= 128, "%s: N=%d unsupported (<= 128 filters per layer)", fn, N);
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
  const size_t smem = cin_fwd_smem(m, H, NT);
  RM_REQUIRE(smem <= 160 * 1024, "rm_cin_layer_fwd: m=%d H=%d needs %zu B of LDS (> 160 KiB)", m, H, smem);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(cin_prep_fwd_kernel, dim3(256), dim3(256), 0, st, W, m, H, N, NT, filter_ws);
  const int epb = kRows / D;
  dim3 grid((unsigned)((B + epb - 1) / epb));
#define RM_CIN_FWD(NT_)                                                                          \
  {                                                                                              \
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(cin_fwd_kernel<NT_>),                     \
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);                  \
    hipLaunchKernelGGL((cin_fwd_kernel<NT_>), grid, dim3(256), smem, st, X0, Xk, xk_bstride,     \
                       filter_ws, bias, act, B, m, H, N, D, out, pooled, pool_stride, pool_col0, \
                       pool_from);                                                               \
  }
  if (NT == 1) RM_CIN_FWD(1) else if (NT == 2) RM_CIN_FWD(2) else RM_CIN_FWD(4)
#undef RM_CIN_FWD
  RM_CHECK_LAUNCH("rm_cin_layer_fwd");
  return RM_OK;
}
