// CIN layer forward (xDeepFM) on the bf16 matrix pipe with split fp32 operands - the scheme of gemm6.hip applied to
//   Z[b,d,i*H+j] = X0[b,i,d] * Xk[b,j,d];  M = Z @ W + bias;  out = act(M) laid out [B,N,D]
// (CIN.__call__, recman/tf/core/layers.py:714-752; csrc/cin.hip is the f32-MFMA form and documents the GEMM view:
// rows p = (b, d), K = m*H, N filters).  Z is formed in registers as the fp32 product the reference forms
// (fl(x0 * xk)), THEN split into three bf16 pieces; the filter is split and laid out in fragment order once per
// call; six exact piece products per k-step, fp32 accumulate: fp32-level error (tests/test_gpu_cin.py).
//
// Covered: H <= 64 (padded to a multiple of 32 with zero filter rows, so that a k' chunk of 32 has ONE i: k' = i * Hp
// + j), N <= 128, D in {16, 32, 64}; anything else returns RM_EUNSUPPORTED and the caller runs rm_cin_layer_fwd.
// The first layer (Xk = X0, H = m) is covered too; whether it is routed here is the caller's choice - cin.hip's
// symmetric k' ordering does half the work there.
//
// Block = 4 waves = 128 rows (128 / D whole examples), two blocks per CU (61 KB of LDS each).  Wave w owns rows
// 32 w .. 32 w + 31 (two 16-row MFMA tiles) x all N / 16 column tiles.  A lane's operand fragment is (row r,
// k = 8 q + e): j = 32 jh + 8 q + e for the slab's (i, jh) - the 8 (16 for H = 64) Xk values a lane EVER needs sit in
// its registers for the whole block; X0's 128 x m tile sits in LDS (one value per row tile and slab).  The filter
// streams by LDS-DMA into two slab buffers, one raw barrier per slab; the next slab's products and split run between
// this slab's MFMAs.  Z is the MFMA's A operand: a lane ends with 4 consecutive d of one filter n - with D = 16 a
// 16 x 16 tile is ONE contiguous KiB of out[b, 16 j .. 16 j + 15, :].  (The same kernel on v_mfma_f32_32x32x16_bf16 -
// half the MFMA instructions for the same matrix work, one row per lane - took 2.20 ms against 2.03 ms.)
#include <type_traits>

#include "rm_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kRowsC6 = 128;   // rows (b, d) per block
constexpr int kNTC6 = 8;       // 16-filter tiles (N <= 128)
constexpr int kSlabC6 = 3 * kNTC6 * 1024;  // bytes of a filter slab: [piece][tile][lane][8] bf16

__device__ __forceinline__ float actc6(float v, int act) {
  if (act == RM_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == RM_ACT_LEAKY_RELU) return v > 0.f ? v : 0.2f * v;
  return v;
}

// Wp[s][p][j][lane][e] = piece p of W[k' = 32 s + 8 (lane >> 4) + e][n = 16 j + (lane & 15)]; two spare slabs behind
// the last (the loop requests them and never reads them)
// (H is padded to Hp = 32 NH: k' = i * Hp + j, zero rows for j >= H)
__global__ void cin6_prep_kernel(const float *__restrict__ W, int H, int Hp, int N, int nslab, __bf16 *__restrict__ Wp) {
  const int64_t total = (int64_t)(nslab + 2) * kNTC6 * 512;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int e = t & 7, lane = (t >> 3) & 63;
    const int64_t u = t >> 9;
    const int j = (int)(u % kNTC6), s = (int)(u / kNTC6);
    const int k = 32 * s + 8 * (lane >> 4) + e, n = 16 * j + (lane & 15);
    const int fi = k / Hp, fj = k - fi * Hp;
    float x = 0.f;
    if (s < nslab && fj < H && n < N) x = W[(int64_t)(fi * H + fj) * N + n];
    const __bf16 h = (__bf16)x;
    const float r1 = x - (float)h;
    const __bf16 m = (__bf16)r1;
    const __bf16 l = (__bf16)(r1 - (float)m);
    const int64_t base = (int64_t)s * (kSlabC6 / 2) + (int64_t)j * 512 + lane * 8 + e;
    Wp[base] = h;
    Wp[base + kNTC6 * 512] = m;
    Wp[base + 2 * kNTC6 * 512] = l;
  }
}

__device__ __forceinline__ void split8c(const float (&x)[8], bf16x8 &h, bf16x8 &m, bf16x8 &l) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const __bf16 hh = (__bf16)x[e];
    const float r1 = x[e] - (float)hh;
    const __bf16 mm = (__bf16)r1;
    const __bf16 ll = (__bf16)(r1 - (float)mm);
    h[e] = hh; m[e] = mm; l[e] = ll;
  }
}

struct Cin6Args {
  const float *X0, *Xk;
  int64_t xk_bstride;
  const __bf16 *Wp;
  const float *bias;
  int act;
  int64_t B;
  int m, H, N, D;
  float *out, *pooled;
  int pool_stride, pool_col0, pool_from;
};

template <int NH>  // H / 32
__global__ __launch_bounds__(256, 2) void cin_fwd6_kernel(Cin6Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smemc6[];
  // [2][kSlabC6] filter slabs | X0s [m][128] floats
  float *X0s = reinterpret_cast<float *>(smemc6 + 2 * kSlabC6);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 15, q = lane >> 4;
  const int D = a.D, m = a.m;
  const int epb = kRowsC6 / D;
  const int64_t b0 = (int64_t)blockIdx.x * epb;
  const int nslab = m * NH;

  auto stage = [&](int s, int buf) {
    const unsigned char *src = reinterpret_cast<const unsigned char *>(a.Wp) + (int64_t)s * kSlabC6 + lane * 16;
    unsigned char *dst = smemc6 + buf * kSlabC6;
#pragma unroll
    for (int i = 0; i < 3 * kNTC6 / 4; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + (4 * i + w) * 1024),
                                       (__attribute__((address_space(3))) void *)(dst + (4 * i + w) * 1024), 16, 0, 0);
  };
  stage(0, 0);

  // X0 tile -> LDS [i][row] (row = local (b, d)): float4 along d
  {
    const int D4 = D >> 2, total = epb * m * D4;
    for (int t = tid; t < total; t += 256) {
      const int d4 = t % D4, i = (t / D4) % m, bl = t / (D4 * m);
      const int64_t b = b0 + bl;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (b < a.B) v = *reinterpret_cast<const float4 *>(a.X0 + (b * m + i) * D + 4 * d4);
      *reinterpret_cast<float4 *>(X0s + i * kRowsC6 + bl * D + 4 * d4) = v;
    }
  }
  // this lane's rows (two tiles) and ALL the Xk values it will ever multiply: j = 32 jh + 8 q + e
  float xk[2][NH][8];
  int prow[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int p = 32 * w + 16 * t + r;
    prow[t] = p;
    const int bl = p / D, d = p - bl * D;
    const int64_t b = b0 + bl < a.B ? b0 + bl : a.B - 1;  // (rows past B: computed, never stored)
    const float *src = a.Xk + b * a.xk_bstride + d;
#pragma unroll
    for (int jh = 0; jh < NH; ++jh)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int j = 32 * jh + 8 * q + e;  // (j >= H: the filter rows are zero, any finite value does)
        xk[t][jh][e] = src[(int64_t)(j < a.H ? j : a.H - 1) * D];
      }
  }

  f32x4 acc[2][kNTC6];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int j = 0; j < kNTC6; ++j) acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();  // X0s and filter slab 0 in place

  // fragments of slab s: z = fl(x0[i] * xk[j]) split into three pieces
  bf16x8 fr[2][6];
  auto make_frags = [&](int s, bf16x8 (&f)[6]) {
    const int sc = s < nslab ? s : nslab - 1;
    const int i = sc / NH, jh = sc - i * NH;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const float x0v = X0s[i * kRowsC6 + prow[t]];
      float z[8];
      if (NH == 1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) z[e] = x0v * xk[t][0][e];
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) z[e] = x0v * (jh ? xk[t][NH - 1][e] : xk[t][0][e]);
      }
      split8c(z, f[3 * t], f[3 * t + 1], f[3 * t + 2]);
    }
  };
  make_frags(0, fr[0]);

  auto slab = [&](int s, auto cur) {
    constexpr int C = decltype(cur)::value, Nx = C ^ 1;
    stage(s + 1, (s + 1) & 1);   // (buffer read during slab s - 1: every wave is past that slab's barrier)
    make_frags(s + 1, fr[Nx]);
    const unsigned char *ws = smemc6 + (s & 1) * kSlabC6 + lane * 16;
#pragma unroll
    for (int j = 0; j < kNTC6; ++j) {
      const bf16x8 bh = *reinterpret_cast<const bf16x8 *>(ws + (0 * kNTC6 + j) * 1024);
      const bf16x8 bm = *reinterpret_cast<const bf16x8 *>(ws + (1 * kNTC6 + j) * 1024);
      const bf16x8 bl = *reinterpret_cast<const bf16x8 *>(ws + (2 * kNTC6 + j) * 1024);
      const bf16x8 ah0 = fr[C][0], am0 = fr[C][1], al0 = fr[C][2], ah1 = fr[C][3], am1 = fr[C][4], al1 = fr[C][5];
      f32x4 c0 = acc[0][j], c1 = acc[1][j];
      // Z is the A operand: D[m = row 4 q + i][n = filter r]; the small products first
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al0, bh, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al1, bh, c1, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah0, bl, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah1, bl, c1, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am0, bm, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am1, bm, c1, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am0, bh, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am1, bh, c1, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah0, bm, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah1, bm, c1, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah0, bh, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah1, bh, c1, 0, 0, 0);
      acc[0][j] = c0;
      acc[1][j] = c1;
    }
    // one split instruction per MFMA gap, the tile's three fragment reads behind its twelve MFMAs (gemm6.hip)
#pragma unroll
    for (int j = 0; j < kNTC6; ++j) {
#pragma unroll
      for (int i = 0; i < 12; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // the next slab's filter pieces have landed
    __builtin_amdgcn_s_barrier();
  };
  int s = 0;
  for (; s + 1 < nslab; s += 2) {
    slab(s, std::integral_constant<int, 0>{});
    slab(s + 1, std::integral_constant<int, 1>{});
  }
  if (s < nslab) slab(s, std::integral_constant<int, 0>{});

  // ---- epilogue: bias + activation, [B, N, D] store, pooled sums ----
  // lane (r, q) holds rows 16 t + 4 q .. + 3 (consecutive d of one example) of filter 16 j + r
  float *pool_s = reinterpret_cast<float *>(smemc6);  // [8 row tiles][128] per-tile sums over their 16 d
  const bool want_pool = a.pooled != nullptr;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int p = 32 * w + 16 * t + 4 * q;
    const int bl = p / D, d = p - bl * D;
    const int64_t b = b0 + bl;
#pragma unroll
    for (int j = 0; j < kNTC6; ++j) {
      const int n = 16 * j + r;
      const float bn = n < a.N ? a.bias[n] : 0.f;
      float4 v;
      v.x = actc6(acc[t][j][0] + bn, a.act);
      v.y = actc6(acc[t][j][1] + bn, a.act);
      v.z = actc6(acc[t][j][2] + bn, a.act);
      v.w = actc6(acc[t][j][3] + bn, a.act);
      if (b < a.B && n < a.N) *reinterpret_cast<float4 *>(a.out + (b * a.N + n) * D + d) = v;
      if (want_pool) {
        // sum over the tile's 16 rows: 4 here, the other 12 in the lanes 16 and 32 away - a fixed order
        float sd = (v.x + v.y) + (v.z + v.w);
        sd += __shfl_xor(sd, 16, 64);
        sd += __shfl_xor(sd, 32, 64);
        if (q == 0) pool_s[(2 * w + t) * 128 + n] = sd;
      }
    }
  }
  if (want_pool) {
    __syncthreads();
    const int ncols = a.N - a.pool_from, tiles = D >> 4;
    for (int t = tid; t < epb * ncols; t += 256) {
      const int bl = t / ncols, cidx = t - bl * ncols;
      const int64_t b = b0 + bl;
      if (b < a.B) {
        float sum = pool_s[(bl * tiles) * 128 + a.pool_from + cidx];
        for (int u = 1; u < tiles; ++u) sum += pool_s[(bl * tiles + u) * 128 + a.pool_from + cidx];
        a.pooled[b * a.pool_stride + a.pool_col0 + cidx] = sum;
      }
    }
  }
}

bool cin6_covers(int m, int H, int N, int D) {
  return H >= 1 && H <= 64 && N <= 128 && (D == 16 || D == 32 || D == 64) && m >= 1 && m <= 64;
}

}  // namespace

// floats of filter workspace for rm_cin_layer_fwd6 (0: the shape is not covered)
extern "C" int64_t rm_cin_filter_workspace6(int m, int H, int N, int D) {
  if (!cin6_covers(m, H, N, D)) return 0;
  const int nslab = m * ((H + 31) / 32);
  return (int64_t)(nslab + 2) * kSlabC6 / 4 + 64;
}

extern "C" int rm_cin_layer_fwd6(const float *X0, const float *Xk, int64_t xk_bstride, const float *W,
                                 const float *bias, int act, int64_t B, int m, int H, int N, int D, float *out,
                                 float *pooled, int pool_stride, int pool_col0, int pool_from, float *filter_ws,
                                 rm_stream_t stream) {
  RM_REQUIRE(B >= 0 && m > 0 && H > 0 && N > 0 && D > 0, "rm_cin_layer_fwd6: bad sizes");
  if (!cin6_covers(m, H, N, D)) {
    rm_set_error("rm_cin_layer_fwd6: shape not covered (H <= 64, N <= 128, D in {16, 32, 64})");
    return RM_EUNSUPPORTED;
  }
  if (B == 0) return RM_OK;
  RM_REQUIRE(X0 && Xk && W && bias && out && filter_ws, "rm_cin_layer_fwd6: NULL argument");
  RM_REQUIRE(rm_aligned16(X0) && rm_aligned16(out) && rm_aligned16(filter_ws), "rm_cin_layer_fwd6: 16-byte alignment required");
  RM_REQUIRE(act >= RM_ACT_IDENTITY && act <= RM_ACT_LEAKY_RELU, "rm_cin_layer_fwd6: bad activation id");
  RM_REQUIRE(!pooled || (pool_from >= 0 && pool_from <= N && pool_stride >= pool_col0 + N - pool_from),
             "rm_cin_layer_fwd6: bad pooled layout");
  hipStream_t st = (hipStream_t)stream;
  const int NH = (H + 31) / 32, nslab = m * NH;
  __bf16 *Wp = reinterpret_cast<__bf16 *>(filter_ws);
  hipLaunchKernelGGL(cin6_prep_kernel, dim3(rm_grid_cap(((int64_t)(nslab + 2) * kNTC6 * 512 + 255) / 256, 2048)),
                     dim3(256), 0, st, W, H, 32 * NH, N, nslab, Wp);
  Cin6Args a{X0, Xk, xk_bstride, Wp, bias, act, B, m, H, N, D, out, pooled, pool_stride, pool_col0, pool_from};
  const int epb = kRowsC6 / D;
  const dim3 grid((unsigned)((B + epb - 1) / epb));
  const size_t smem = 2 * kSlabC6 + (size_t)m * kRowsC6 * 4;
  if (NH == 1) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(cin_fwd6_kernel<1>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(cin_fwd6_kernel<1>, grid, dim3(256), smem, st, a);
  } else {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(cin_fwd6_kernel<2>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(cin_fwd6_kernel<2>, grid, dim3(256), smem, st, a);
  }
  RM_CHECK_LAUNCH("rm_cin_layer_fwd6");
  return RM_OK;
}

// ---- backward, the dX part (cin.hip's cin_dx_kernel on the same scheme) ---------------------------------------
//   dZ[p, (i, j)] = sum_n W[(i, j), n] dM[p, n];  dX0[b, i, d] += sum_j dZ Xk[b, j, d];  dXk[b, j, d] += sum_i dZ X0[b, i, d]
// dZ^T tile [16 k'][16 rows] = W[k'][:] . dM[row][:]^T with the FILTER as the MFMA's A operand (split once per call:
// cin6_prep_dx_kernel) and the lane's own dM rows as the B operand - split ONCE per block into 3 x 16 bf16 fragments
// per row tile and kept in registers (K = N = 128 = four 32-n slabs).  A lane ends with dZ of 4 consecutive j of one
// row: the contraction with Xk / X0 runs in registers (8 FMAs per 24 MFMAs), the Xk values and the dXk accumulators
// of the lane's (row, j) pairs live in registers for the whole block, dX0 goes through LDS.  Filter slabs (one i,
// 32 j, all n: 24 KiB) stream by LDS-DMA into two buffers; 4-wave blocks of 128 rows, two per CU.
namespace {

// Wq6[s = i * NH + jh][kt][ns][p][lane][e] = piece p of W[(i * H + 32 jh + 16 kt + (lane & 15)) * N + 32 ns + 8 (lane >> 4) + e]
__global__ void cin6_prep_dx_kernel(const float *__restrict__ W, int m, int H, int N, __bf16 *__restrict__ Wp) {
  const int NH = (H + 31) / 32, nslab = m * NH;
  const int64_t total = (int64_t)(nslab + 2) * 8 * 512;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int e = t & 7, lane = (t >> 3) & 63;
    const int64_t u = t >> 9;
    const int ns = (int)(u & 3), kt = (int)((u >> 2) & 1), s = (int)(u >> 3);
    float x = 0.f;
    if (s < nslab) {
      const int i = s / NH, jh = s - i * NH;
      const int j = 32 * jh + 16 * kt + (lane & 15), n = 32 * ns + 8 * (lane >> 4) + e;
      if (n < N && j < H) x = W[(int64_t)(i * H + j) * N + n];
    }
    const __bf16 h = (__bf16)x;
    const float r1 = x - (float)h;
    const __bf16 mm = (__bf16)r1;
    const __bf16 l = (__bf16)(r1 - (float)mm);
    const int64_t base = (int64_t)s * (kSlabC6 / 2) + (int64_t)((kt * 4 + ns) * 3) * 512 + lane * 8 + e;
    Wp[base] = h;
    Wp[base + 512] = mm;
    Wp[base + 1024] = l;
  }
}

struct CinDx6Args {
  const float *X0, *Xk;
  int64_t xk_bstride;
  const __bf16 *Wp;
  const float *dM;  // [B * D][128]
  int64_t B;
  int m, H, D;
  float *dX0;
  int accumulate_dx0, xk_is_x0;
  float *dXk;
  int64_t dxk_bstride;
};

template <int NH>
__global__ __launch_bounds__(256, 2) void cin_dx6_kernel(CinDx6Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smemc6[];
  // [2][kSlabC6] filter slabs (later: the dXk image [H][128]) | X0s [m][128] | dX0s [m][128]
  float *X0s = reinterpret_cast<float *>(smemc6 + 2 * kSlabC6);
  float *dX0s = X0s + a.m * kRowsC6;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 15, q = lane >> 4;
  const int D = a.D, m = a.m;
  const int epb = kRowsC6 / D;
  const int64_t b0 = (int64_t)blockIdx.x * epb;
  constexpr int JT = 2 * NH;  // 16-j tiles

  auto stage = [&](int s, int buf) {
    const unsigned char *src = reinterpret_cast<const unsigned char *>(a.Wp) + (int64_t)s * kSlabC6 + lane * 16;
    unsigned char *dst = smemc6 + buf * kSlabC6;
#pragma unroll
    for (int i = 0; i < 6; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + (4 * i + w) * 1024),
                                       (__attribute__((address_space(3))) void *)(dst + (4 * i + w) * 1024), 16, 0, 0);
  };
  stage(0, 0);
  {
    const int D4 = D >> 2, total = epb * m * D4;
    for (int t = tid; t < total; t += 256) {
      const int d4 = t % D4, i = (t / D4) % m, bl = t / (D4 * m);
      const int64_t b = b0 + bl;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (b < a.B) v = *reinterpret_cast<const float4 *>(a.X0 + (b * m + i) * D + 4 * d4);
      *reinterpret_cast<float4 *>(X0s + i * kRowsC6 + bl * D + 4 * d4) = v;
    }
  }
  // the lane's rows: dM fragments (split once), the Xk values of its (row, j = 16 jt + 4 q + i') pairs, dXk sums
  bf16x8 dmf[2][4][3];
  float xkr[2][JT][4], dxk[2][JT][4];
  int prow[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int p = 32 * w + 16 * t + r;
    prow[t] = p;
    const int64_t pg = b0 * D + p;
    const bool live = pg < a.B * D;
    const float *src = a.dM + (live ? pg : 0) * 128 + 8 * q;
#pragma unroll
    for (int ns = 0; ns < 4; ++ns) {
      const float4 u = *reinterpret_cast<const float4 *>(src + 32 * ns), v = *reinterpret_cast<const float4 *>(src + 32 * ns + 4);
      float y[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) y[e] = live ? y[e] : 0.f;
      split8c(y, dmf[t][ns][0], dmf[t][ns][1], dmf[t][ns][2]);
    }
    const int bl = p / D, d = p - bl * D;
    const int64_t b = b0 + bl < a.B ? b0 + bl : a.B - 1;
    const float *xs = a.Xk + b * a.xk_bstride + d;
#pragma unroll
    for (int jt = 0; jt < JT; ++jt)
#pragma unroll
      for (int i2 = 0; i2 < 4; ++i2) {
        const int j = 16 * jt + 4 * q + i2;  // (j >= H: dZ is zero there - zero filter rows)
        xkr[t][jt][i2] = xs[(int64_t)(j < a.H ? j : a.H - 1) * D];
        dxk[t][jt][i2] = 0.f;
      }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();

  for (int i = 0; i < m; ++i) {
    const float x0v[2] = {X0s[i * kRowsC6 + prow[0]], X0s[i * kRowsC6 + prow[1]]};
    float dx0[2] = {0.f, 0.f};
#pragma unroll
    for (int jh = 0; jh < NH; ++jh) {
      const int s = i * NH + jh;
      stage(s + 1, (s + 1) & 1);  // (two spare slabs behind the last)
      const unsigned char *ws = smemc6 + (s & 1) * kSlabC6 + lane * 16;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ns = 0; ns < 4; ++ns) {
          const unsigned char *wp = ws + ((kt * 4 + ns) * 3) * 1024;
          const bf16x8 wh = *reinterpret_cast<const bf16x8 *>(wp);
          const bf16x8 wm = *reinterpret_cast<const bf16x8 *>(wp + 1024);
          const bf16x8 wl = *reinterpret_cast<const bf16x8 *>(wp + 2048);
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            f32x4 c = acc[t];
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, dmf[t][ns][0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, dmf[t][ns][2], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, dmf[t][ns][1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, dmf[t][ns][0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, dmf[t][ns][1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, dmf[t][ns][0], c, 0, 0, 0);
            acc[t] = c;
          }
        }
        // acc[t][i2] = dZ[row prow[t]][(i, j = 32 jh + 16 kt + 4 q + i2)]
        const int jt = 2 * jh + kt;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int i2 = 0; i2 < 4; ++i2) {
            dx0[t] += acc[t][i2] * xkr[t][jt][i2];
            dxk[t][jt][i2] += acc[t][i2] * x0v[t];
          }
      }
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float v = dx0[t];
      v += __shfl_xor(v, 16, 64);  // the four lane quarters hold disjoint j
      v += __shfl_xor(v, 32, 64);
      if (q == 0) dX0s[i * kRowsC6 + prow[t]] = v;
    }
  }
  // dXk -> LDS image [j][row] over the filter buffers (every (j, row) is owned by exactly one lane)
  float *dXks = reinterpret_cast<float *>(smemc6);
  if (a.dXk != nullptr || a.xk_is_x0) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int jt = 0; jt < JT; ++jt)
#pragma unroll
        for (int i2 = 0; i2 < 4; ++i2) dXks[(16 * jt + 4 * q + i2) * kRowsC6 + prow[t]] = dxk[t][jt][i2];
  }
  __syncthreads();
  const int D4 = D >> 2;
  for (int t = tid; t < epb * m * D4; t += 256) {
    const int d4 = t % D4, i = (t / D4) % m, bl = t / (D4 * m);
    const int64_t b = b0 + bl;
    if (b >= a.B) continue;
    float4 v = *reinterpret_cast<const float4 *>(dX0s + i * kRowsC6 + bl * D + 4 * d4);
    if (a.xk_is_x0) {  // first layer: Xk IS X0 (H = m), its gradient lands on the same field
      const float4 w4 = *reinterpret_cast<const float4 *>(dXks + i * kRowsC6 + bl * D + 4 * d4);
      v.x += w4.x; v.y += w4.y; v.z += w4.z; v.w += w4.w;
    }
    float4 *dst = reinterpret_cast<float4 *>(a.dX0 + (b * m + i) * D + 4 * d4);
    if (a.accumulate_dx0) {
      const float4 o = *dst;
      v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
    }
    *dst = v;
  }
  if (a.dXk != nullptr && !a.xk_is_x0) {
    const int H = a.H;
    for (int t = tid; t < epb * H * D4; t += 256) {
      const int d4 = t % D4, j = (t / D4) % H, bl = t / (D4 * H);
      const int64_t b = b0 + bl;
      if (b >= a.B) continue;
      *reinterpret_cast<float4 *>(a.dXk + b * a.dxk_bstride + (int64_t)j * D + 4 * d4) =
          *reinterpret_cast<const float4 *>(dXks + j * kRowsC6 + bl * D + 4 * d4);
    }
  }
}

}  // namespace

// floats rm_cin_layer_bwd's workspace needs for the split filter of the dX kernel (0: not covered)
int64_t rm_internal_cin_dx6_floats(int m, int H, int N, int D) {
  if (!cin6_covers(m, H, N, D) || N <= 64) return 0;
  return (int64_t)(m * ((H + 31) / 32) + 2) * kSlabC6 / 4 + 64;
}

// the dX part of rm_cin_layer_bwd on the bf16 pipe: returns false when the shape is not covered (nothing launched)
bool rm_internal_cin_dx6(const float *X0, const float *Xk, int64_t xk_bstride, int xk_is_x0, const float *W,
                         const float *dM, int64_t B, int m, int H, int N, int D, float *dX0, int accumulate_dx0,
                         float *dXk, int64_t dxk_bstride, float *ws6, hipStream_t st) {
  if (rm_internal_cin_dx6_floats(m, H, N, D) == 0 || !rm_aligned16(ws6)) return false;
  const int NH = (H + 31) / 32, nslab = m * NH;
  __bf16 *Wp = reinterpret_cast<__bf16 *>(ws6);
  hipLaunchKernelGGL(cin6_prep_dx_kernel, dim3(rm_grid_cap(((int64_t)(nslab + 2) * 8 * 512 + 255) / 256, 2048)),
                     dim3(256), 0, st, W, m, H, N, Wp);
  CinDx6Args a{X0, Xk, xk_bstride, Wp, dM, B, m, H, D, dX0, accumulate_dx0, xk_is_x0, dXk, dxk_bstride};
  const int epb = kRowsC6 / D;
  const dim3 grid((unsigned)((B + epb - 1) / epb));
  const size_t smem = 2 * kSlabC6 + (size_t)2 * m * kRowsC6 * 4;
  if (NH == 1) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(cin_dx6_kernel<1>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(cin_dx6_kernel<1>, grid, dim3(256), smem, st, a);
  } else {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(cin_dx6_kernel<2>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(cin_dx6_kernel<2>, grid, dim3(256), smem, st, a);
  }
  return true;
}

// ---- backward, the dW part:  dW[(i, j), n] = sum over rows p = (b, d) of Z[p, (i, j)] dM[p, n] ---------------------
// A reduction over the 1 M rows: a slab is 32 ROWS.  An operand fragment is 8 consecutive rows of one column:
//   Z   8 consecutive rows = 8 consecutive d of one example: x0[i] and xk[j] are each 32 contiguous bytes of X0 / Xk -
//       loaded straight from global memory (a slab ahead), multiplied in fp32, split in registers;
//   dM  strided in its row-major [rows][128] image: a slab goes through LDS as transposed bf16 planes
//       [piece][n][32 rows] (a thread loads one column's 8 rows - coalesced across the wave's 64 columns - splits
//       them, writes three rotated 16-byte pieces; gemm6.hip's TN kernel), double-buffered: one barrier per slab.
// Block = 4 waves = 3 fields i x all 64 j x all 128 n (wave w = the 16 j of tile w for the block's 3 i: 3 x 8
// accumulator tiles), two blocks per CU; dM is read once per group of 3 fields (9 times at m = 26: 4.8 GB per layer,
// streaming).  dM is the MFMA's A operand: a lane ends with 4 consecutive n of one (i, j).  Partial sums per row split
// go to the workspace; cin6_dw_reduce_kernel adds them in split order (deterministic).
namespace {

#ifndef RM_CIN_DW6_WAVES
#define RM_CIN_DW6_WAVES 4  // (8: dM staged half as often, but one block per CU - 2.64 ms against 2.54 ms)
#endif
#ifndef RM_CIN_DW6_ABL
#define RM_CIN_DW6_ABL 0  // ablation builds (WRONG results): 1 no MFMAs, 2 no dM loads / planes in the loop, 4 no Z loads / split in the loop
#endif
constexpr int kDwI = 3;                    // fields per wave (4: 256 VGPRs + 100 B of scratch, 5.3 ms against 4.9 for the layer backward)
constexpr int kDwWaves = RM_CIN_DW6_WAVES; // 4: one block = 3 fields, two blocks per CU; 8: one block = 6 fields (waves
                                           // 4 .. 7 take the second three), one per CU - dM staged and split half as often
constexpr int kDwFields = kDwI * (kDwWaves / 4);
constexpr int kDwUnits = 512 / (64 * kDwWaves);  // dM units (column, 8-row group) per thread and slab
constexpr int kDwPlane = 128 * 64;         // bytes of one bf16 plane [128 n][32 rows]
constexpr int kDwBuf = 3 * kDwPlane;       // 24 KiB

struct CinDw6Args {
  const float *X0, *Xk, *dM;
  int64_t xk_bstride;
  int64_t B;
  int m, H, D;
  int ngroups, nsplit;
  int64_t slabs_per_split, nslab;
  float *part;  // [nsplit][m * H][128]
};

template <int NH>
__global__ __launch_bounds__(64 * kDwWaves, kDwWaves == 4 ? 2 : 1) void cin_dw6_kernel(CinDw6Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smemc6[];  // [2][kDwBuf]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, w = wv & 3, r = lane & 15, q = lane >> 4;
  const int D = a.D, H = 32 * NH;  // (the padded width; a.H is the real one)
  // blockIdx -> (field group, row split): the field groups of ONE row split - they stream the same dM rows at about
  // the same pace - are 8 blocks apart, i.e. on the same XCD (consecutive workgroups go round-robin to the 8 XCDs,
  // each with its own L2): dM comes from HBM about once instead of once per field group
  int grp, sp;
  {
    const int per = 8 * a.ngroups, sg = blockIdx.x / per, in = blockIdx.x % per;
    if ((sg + 1) * 8 <= a.nsplit) { grp = in / 8; sp = sg * 8 + in % 8; }
    else { const int rest = blockIdx.x - sg * per; grp = rest % a.ngroups; sp = sg * 8 + rest / a.ngroups; }
  }
  const int i0 = kDwFields * grp + kDwI * (wv >> 2);
  const int64_t rows_total = a.B * D;
  constexpr int JW = NH * 2 / 4 > 0 ? NH * 2 / 4 : 1;  // 16-j tiles per wave: H = 64 -> 1, H = 32 -> waves 2, 3 idle
  const bool wave_on = 16 * w < H && i0 < a.m;

  // dM units of this thread: (column n, 8-row group g), two per slab
  int un[kDwUnits], ug[kDwUnits], ulds[kDwUnits];
#pragma unroll
  for (int k = 0; k < kDwUnits; ++k) {
    const int u = tid + 64 * kDwWaves * k;
    un[k] = u & 127;
    ug[k] = u >> 7;
    ulds[k] = un[k] * 64 + ((ug[k] + (un[k] >> 2)) & 3) * 16;  // (pieces rotated by n / 4: gemm6.hip)
  }
  auto load_dm = [&](int64_t slab, float (&x)[kDwUnits][8]) {
#pragma unroll
    for (int k = 0; k < kDwUnits; ++k)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int64_t row = slab * 32 + 8 * ug[k] + e;
        x[k][e] = a.dM[(row < rows_total ? row : rows_total - 1) * 128 + un[k]];
      }
  };
  // Z sources of this lane: rows 8 q .. 8 q + 7 of the slab = 8 consecutive d of one example
  auto load_z = [&](int64_t slab, float (&xkv)[8], float (&x0v)[kDwI][8]) {
    int64_t row = slab * 32 + 8 * q;
    row = row < rows_total ? row : rows_total - 8;  // (rows past the end: dM is zeroed there, any finite value does)
    const int64_t b = row / D;
    const int d = (int)(row - b * D);
    int j = 16 * (wave_on ? w : 0) + r;
    j = j < a.H ? j : a.H - 1;  // (padded j: its dW rows are dropped by the reduction)
    const float4 *pk = reinterpret_cast<const float4 *>(a.Xk + b * a.xk_bstride + (int64_t)j * D + d);
    const float4 k0 = pk[0], k1 = pk[1];
    xkv[0] = k0.x; xkv[1] = k0.y; xkv[2] = k0.z; xkv[3] = k0.w; xkv[4] = k1.x; xkv[5] = k1.y; xkv[6] = k1.z; xkv[7] = k1.w;
#pragma unroll
    for (int u = 0; u < kDwI; ++u) {
      const int i = i0 + u < a.m ? i0 + u : a.m - 1;
      const float4 *p0 = reinterpret_cast<const float4 *>(a.X0 + (b * a.m + i) * D + d);
      const float4 v0 = p0[0], v1 = p0[1];
      x0v[u][0] = v0.x; x0v[u][1] = v0.y; x0v[u][2] = v0.z; x0v[u][3] = v0.w;
      x0v[u][4] = v1.x; x0v[u][5] = v1.y; x0v[u][6] = v1.z; x0v[u][7] = v1.w;
    }
  };
  auto write_dm = [&](int64_t slab, const float (&x)[kDwUnits][8], int buf) {
#pragma unroll
    for (int k = 0; k < kDwUnits; ++k) {
      float y[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) y[e] = (slab * 32 + 8 * ug[k] + e < rows_total) ? x[k][e] : 0.f;
      bf16x8 h, mm, l;
      split8c(y, h, mm, l);
      unsigned char *dst = smemc6 + buf * kDwBuf + ulds[k];
      *reinterpret_cast<bf16x8 *>(dst) = h;
      *reinterpret_cast<bf16x8 *>(dst + kDwPlane) = mm;
      *reinterpret_cast<bf16x8 *>(dst + 2 * kDwPlane) = l;
    }
  };

  f32x4 acc[kDwI][8];
#pragma unroll
  for (int u = 0; u < kDwI; ++u)
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) acc[u][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int64_t s0 = (int64_t)sp * a.slabs_per_split;
  const int64_t s1 = s0 + a.slabs_per_split < a.nslab ? s0 + a.slabs_per_split : a.nslab;
  float dmr[kDwUnits][8], xkv[8], x0v[kDwI][8];
  if (s0 < s1) {
    load_dm(s0, dmr);
    load_z(s0, xkv, x0v);
    write_dm(s0, dmr, 0);
    load_dm(s0 + 1 < s1 ? s0 + 1 : s0, dmr);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int64_t s = s0; s < s1; ++s) {
    const int buf = (int)((s - s0) & 1);
    // this slab's Z fragments (values loaded a slab ago), then the next slab's on their way
    bf16x8 zf[kDwI][3];
    if (!(RM_CIN_DW6_ABL & 4) || s == s0) {
#pragma unroll
      for (int u = 0; u < kDwI; ++u) {
        float z[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) z[e] = x0v[u][e] * xkv[e];
        split8c(z, zf[u][0], zf[u][1], zf[u][2]);
      }
    }
    const int64_t sn = s + 1 < s1 ? s + 1 : s;
    if (!(RM_CIN_DW6_ABL & 4)) load_z(sn, xkv, x0v);
    // the next slab's dM planes into the other buffer (its raw values were requested a slab ago), the one after
    if (!(RM_CIN_DW6_ABL & 2)) {
      write_dm(sn, dmr, buf ^ 1);
      load_dm(s + 2 < s1 ? s + 2 : sn, dmr);
    }
    if (wave_on) {
      const unsigned char *ws = smemc6 + buf * kDwBuf;
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
        const int n = 16 * nt + r;
        const unsigned char *p = ws + n * 64 + ((q + (n >> 2)) & 3) * 16;
        const bf16x8 gh = *reinterpret_cast<const bf16x8 *>(p);
        const bf16x8 gm = *reinterpret_cast<const bf16x8 *>(p + kDwPlane);
        const bf16x8 gl = *reinterpret_cast<const bf16x8 *>(p + 2 * kDwPlane);
#pragma unroll
        for (int u = 0; u < kDwI; ++u) {
          // dM is the A operand: D[m = n 4 q + i'][col = j r]
          f32x4 c = acc[u][nt];
          if (RM_CIN_DW6_ABL & 1) {
            c[0] += (float)gl[0] + (float)gm[1] + (float)gh[2] + (float)zf[u][0][3] + (float)zf[u][1][4] + (float)zf[u][2][5];
            acc[u][nt] = c;
            continue;
          }
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gl, zf[u][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gh, zf[u][2], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gm, zf[u][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gm, zf[u][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gh, zf[u][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gh, zf[u][0], c, 0, 0, 0);
          acc[u][nt] = c;
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  // partial tile: lane (r, q) holds n = 16 nt + 4 q .. + 3 of k' = (i0 + u) * H + 16 w + r
  if (wave_on) {
    float *pt = a.part + (int64_t)sp * a.m * H * 128;
#pragma unroll
    for (int u = 0; u < kDwI; ++u) {
      if (i0 + u >= a.m) continue;
      const int64_t kp = (int64_t)(i0 + u) * H + 16 * w + r;
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) *reinterpret_cast<f32x4 *>(pt + kp * 128 + 16 * nt + 4 * q) = acc[u][nt];
    }
  }
  (void)JW;
}

// dW[(i * H + j), n] = the partial sums of row i * Hp + j in split order
__global__ void cin6_dw_reduce_kernel(const float *__restrict__ part, int m, int H, int Hp, int N, int nsplit,
                                      float *__restrict__ dW) {
  const int64_t total = (int64_t)m * H * N;
  const int64_t Kp = (int64_t)m * Hp;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t k = t / N;
    const int n = (int)(t - k * N);
    const int i = (int)(k / H), j = (int)(k - (int64_t)i * H);
    const float *p = part + ((int64_t)i * Hp + j) * 128 + n;
    float v = 0.f;
    for (int s = 0; s < nsplit; ++s) v += p[(int64_t)s * Kp * 128];
    dW[t] = v;
  }
}

struct Dw6Plan {
  int ngroups, nsplit;
  int64_t nslab, per;
};
Dw6Plan dw6_plan(int64_t B, int m, int D) {
  Dw6Plan p;
  p.ngroups = (m + kDwFields - 1) / kDwFields;
  p.nslab = (B * D + 31) / 32;
  int64_t want = (kDwWaves == 4 ? 512 : 256) / p.ngroups;  // one round of blocks
  if (want < 1) want = 1;
  if (want > p.nslab) want = p.nslab;
  p.per = (p.nslab + want - 1) / want;
  p.nsplit = (int)((p.nslab + p.per - 1) / p.per);
  return p;
}

}  // namespace

// floats of workspace for the dW part (0: not covered)
int64_t rm_internal_cin_dw6_floats(int64_t B, int m, int H, int N, int D) {
  if (!cin6_covers(m, H, N, D) || N <= 64 || B <= 0) return 0;
  const Dw6Plan p = dw6_plan(B, m, D);
  return (int64_t)p.nsplit * m * (32 * ((H + 31) / 32)) * 128 + 64;
}

bool rm_internal_cin_dw6(const float *X0, const float *Xk, int64_t xk_bstride, const float *dM, int64_t B, int m, int H,
                         int N, int D, float *dW, float *ws6, hipStream_t st) {
  if (rm_internal_cin_dw6_floats(B, m, H, N, D) == 0 || !rm_aligned16(ws6) || !rm_aligned16(X0) || !rm_aligned16(Xk) ||
      xk_bstride % 4 != 0)
    return false;
  const Dw6Plan p = dw6_plan(B, m, D);
  CinDw6Args a{X0, Xk, dM, xk_bstride, B, m, H, D, p.ngroups, p.nsplit, p.per, p.nslab, ws6};
  const dim3 grid((unsigned)(p.ngroups * p.nsplit));
  const size_t smem = 2 * kDwBuf;
  if (H <= 32) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(cin_dw6_kernel<1>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(cin_dw6_kernel<1>, grid, dim3(64 * kDwWaves), smem, st, a);
  } else {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(cin_dw6_kernel<2>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(cin_dw6_kernel<2>, grid, dim3(64 * kDwWaves), smem, st, a);
  }
  hipLaunchKernelGGL(cin6_dw_reduce_kernel, dim3(rm_grid_cap(((int64_t)m * H * N + 255) / 256, 2048)), dim3(256), 0, st,
                     ws6, m, H, 32 * ((H + 31) / 32), N, p.nsplit, dW);
  return true;
}
