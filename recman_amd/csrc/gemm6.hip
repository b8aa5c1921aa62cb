// Wide dense layers with fp32 operands SPLIT into three bf16 pieces each and the products formed on the bf16
// matrix pipe (v_mfma_f32_16x16x32_bf16), fp32 accumulate: the "NN" GEMM C[M,N] = epilogue([A1 | A2][M,K] . op(W))
// (rm_dense_fwd6, first half of this file) and the "TN" weight gradient dW[K,N] = [A1 | A2]^T . G (rm_dense_wgrad6,
// second half).
//
//   x = h + m + l,  h = bf16(x), m = bf16(x - h), l = bf16(x - h - m)   (8 + 8 + 8 = the 24 significant bits)
//   x y ~= hx hy + hx my + mx hy + mx my + hx ly + lx hy               (dropped: m l, l m, l l <= 3 * 2^-24 |x y|)
//
// Every kept product of two bf16 values is EXACT in fp32 (16 significant bits), the sum runs in fp32 as on the
// f32 MFMA: the result carries fp32-level rounding error (tests/test_gpu_dense.py holds it to the same bound
// against float64 as the f32-MFMA kernel of gemm.hip; it is not bit-identical to it - neither kernel is to the
// CPU oracle, whose sums run in another order).  Why: gfx950 has no tf32 / xf32; the f32 MFMA runs at 1/16 of
// the bf16 rate, so six bf16 MFMAs per k-step are 2.7x the f32 pipe's speed-of-light.  gemm.hip's NN kernel
// sits at 0.74-0.80 of THAT pipe (DCN configs[3]: 4 launches x 355-382 us of a 2.85 ms step).
//
// Replaces tf.matmul + bias + activation of DNN.__call__ (recman/tf/core/layers.py:594-602) and the
// data-gradient GEMM of its backward, like rm_dense_fwd's f32 path.
//
// Block = 4 waves (one per SIMD) = 128 batch rows x ONE column group of NT <= 13 16-column tiles, TWO blocks per CU
// (80 KB of LDS each): the blocks drift out of phase, so one's MFMAs cover the other's splitting, staging, barrier
// and epilogue (8 waves in ONE block ran every phase at the same time on both waves of a SIMD).  Wave w owns rows
// [32 w, 32 w + 32) (two 16-row MFMA tiles) and all NT column tiles: 2 x NT x 4 accumulator registers.
//   A   never touches LDS: a lane's operand fragment is 8 consecutive k of one row = 32 contiguous bytes of the
//       row-major activations, loaded straight from global memory three k-slabs (of 32 k) ahead and split in
//       registers between the MFMAs of the slab before its use.
//   W   is split and laid out in fragment order once per call by dense6_prep_kernel (bf16 [group][slab][piece]
//       [tile][lane][8]); a block copies its group's 3 x NT KiB per slab into LDS with LDS-DMA (a linear image:
//       lane l of a fragment read takes bytes 16 l .. 16 l + 15, conflict-free), double-buffered, one barrier
//       per slab.
#include <cstdlib>
#include <type_traits>

#include "rm_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// ablation builds (WRONG results; tools/probe/nn6_time.py): 1 no epilogue stores, 2 no A loads in the loop,
// 4 no splitting in the loop, 8 one B fragment triple per slab, 16 no staging in the loop, 32 no MFMAs,
// 64 no sched_group_barrier pipeline
#ifndef RM_NN6_ABL
#define RM_NN6_ABL 0
#endif
#ifdef RM_NN6_STAMP
__device__ unsigned long long rm_nn6_stamp_buf[4 * 4096];
#endif
#ifndef RM_NN6_WAVES
#define RM_NN6_WAVES 4
#endif
constexpr int kWaves6 = RM_NN6_WAVES;       // 4: one wave per SIMD and TWO blocks per CU (80 KB of LDS each)
constexpr int kRows6 = 32 * kWaves6;        // batch rows per block
constexpr int kThreads6 = 64 * kWaves6;
constexpr int kRing6 = kWaves6 == 4 ? 2 : 3;  // LDS slab buffers
constexpr int kSlabK = 32;      // k per slab = one 16x16x32 MFMA step
constexpr int kMaxNT6 = 13;     // 16-column tiles per column group (N = 400 / 416: two groups of 13)

struct NN6Args {
  const float *A1, *Atail;  // A1 [M, >= 32 nfull] (16-byte aligned rows); the padded ragged end [M, 32] or NULL
  int64_t lda1;
  int nfull;         // whole slabs inside A1
  int tail_k0;       // Atail == NULL and a ragged K: the last slab re-reads A1's columns [tail_k0, tail_k0 + 32)
  const float *dot_w;  // or NULL: also dot_part[g * M + row] = sum over the group's columns of C[row, col] * dot_w[col]
  float *dot_part;
  const __bf16 *Wp;  // prepped weights
  int N, nslab, ngroups;
  const float *bias;
  int epi, act;
  const float *aux1;
  int64_t ld1;
  int64_t M;
  float *C;
  int64_t ldc;
};

__device__ __forceinline__ float act6(float v, int act) {
  if (act == RM_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == RM_ACT_LEAKY_RELU) return v > 0.f ? v : 0.2f * v;
  return v;
}
__device__ __forceinline__ float actgrad6(float o, int act) {
  if (act == RM_ACT_RELU) return o > 0.f ? 1.f : 0.f;
  if (act == RM_ACT_LEAKY_RELU) return o > 0.f ? 1.f : 0.2f;
  return 1.f;
}

// Wp[g][s][p][j][lane][e] = piece p of op(W)[k = 32 s + 8 (lane >> 4) + e][col = 16 (NT g + j) + (lane & 15)]
template <int NT>
__global__ void dense6_prep_kernel(const float *__restrict__ W, int64_t ldw, int trans, int K, int N, int nslab,
                                   int ngroups, int overlap_k0, __bf16 *__restrict__ Wp) {
  const int64_t total = (int64_t)ngroups * nslab * NT * 64 * 8;  // elements per piece
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int e = t & 7, lane = (t >> 3) & 63;
    const int64_t u = t >> 9;
    const int j = (int)(u % NT);
    const int s = (int)((u / NT) % nslab), g = (int)(u / NT / nslab);
    int k = kSlabK * s + 8 * (lane >> 4) + e;
    const int col = 16 * (NT * g + j) + (lane & 15);
    bool live = k < K;
    if (overlap_k0 >= 0 && s == nslab - 1) {
      // the last slab of a ragged K re-reads the activations' columns [K - 32, K): the weights of those the
      // slab before has already covered are zero here
      k = overlap_k0 + 8 * (lane >> 4) + e;
      live = k >= kSlabK * (nslab - 1);
    }
    float x = 0.f;
    if (live && col < N) x = trans ? W[(int64_t)col * ldw + k] : W[(int64_t)k * ldw + col];
    const __bf16 h = (__bf16)x;
    const float r1 = x - (float)h;
    const __bf16 m = (__bf16)r1;
    const __bf16 l = (__bf16)(r1 - (float)m);
    constexpr int kPieces = (3 * NT + 7) / 8 * 8;  // (padded: every wave issues the same number of DMA pieces)
    const int64_t base = (int64_t)(g * (nslab + 2) + s) * kPieces * 512 + (int64_t)j * 512 + lane * 8 + e;
    Wp[base] = h;
    Wp[base + (int64_t)NT * 512] = m;
    Wp[base + (int64_t)2 * NT * 512] = l;
  }
}

// The ragged end of [A1 | A2] - A1's columns past its last whole slab and all of A2 - copied into Atail [M, 32]
// (zero padded): every slab of the main kernel is then two unconditional float4 loads per row.  (A per-lane branch
// around a load costs the overlap: hipcc answers it with s_waitcnt vmcnt(0) at the join or sinks the load to its
// use.)  13 dense columns at DCN's layer 0: 17 MB written and read, ~8 us of a 300 us launch.
__global__ void dense6_tail_kernel(const float *__restrict__ A1, int64_t lda1, int K1, const float *__restrict__ A2,
                                   int64_t lda2, int K2, int kfull, int64_t M, float *__restrict__ Atail) {
  const int64_t total = M * kSlabK;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = t / kSlabK;
    const int k = kfull + (int)(t % kSlabK);
    float v = 0.f;
    if (k < K1) v = A1[row * lda1 + k];
    else if (k < K1 + K2) v = A2[row * lda2 + (k - K1)];
    Atail[t] = v;
  }
}

__device__ __forceinline__ void load_a8(const float *base, int64_t ld, int64_t row, int k0, float (&x)[8]) {
  const float4 *p = reinterpret_cast<const float4 *>(base + row * ld + k0);
  const float4 u = p[0], v = p[1];
  x[0] = u.x; x[1] = u.y; x[2] = u.z; x[3] = u.w; x[4] = v.x; x[5] = v.y; x[6] = v.z; x[7] = v.w;
}

__device__ __forceinline__ void split8(const float (&x)[8], bf16x8 &h, bf16x8 &m, bf16x8 &l) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const __bf16 hh = (__bf16)x[e];
    const float r1 = x[e] - (float)hh;
    const __bf16 mm = (__bf16)r1;
    const __bf16 ll = (__bf16)(r1 - (float)mm);
    h[e] = hh; m[e] = mm; l[e] = ll;
  }
}

template <int NT>
__global__ __launch_bounds__(kThreads6, kWaves6 == 4 ? 2 : 1) void dense_nn6_kernel(NN6Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem6[];
  constexpr int kPieces = (3 * NT + 7) / 8 * 8;  // 1 KiB pieces per slab in GLOBAL memory, padded to 8 per DMA round
  constexpr int kSlabBytes = kPieces * 1024;
  constexpr int kLdsSlab = 3 * NT * 1024;        // in LDS only the real pieces have a slot; the padding goes to a sink
  constexpr int kSink = kRing6 * kLdsSlab;       // (2 x (2 x 39 + 1) KiB = 158 KiB: two blocks per CU)
  constexpr int kDmaPerWave = kPieces / kWaves6;  // LDS-DMA pieces a wave issues per slab
  constexpr int kAhead = kRing6 - 1;              // slabs the weights run ahead
  // a slab ends when the NEXT slab's weights have landed: with a ring of 3 everything older than this slab's own
  // requests, with a ring of 2 this slab's own LDS-DMA pieces too (only the 4 row loads behind them may remain)
  constexpr int kVmPerSlab = kRing6 == 3 ? kDmaPerWave + 4 : 4;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  // blockIdx -> (row tile, column group): consecutive workgroups go round-robin to the 8 XCDs (each its own L2);
  // the column groups of ONE row tile are 8 blocks apart - the same XCD, dispatched together - so that the second
  // reader of the A tile finds it in that L2 (tile * ngroups + g put them on two XCDs: 711 MB of HBM traffic for
  // 436 MB of operands).  Super-groups of 8 tiles x ngroups; a ragged last super-group falls back to the plain map.
  int g;
  int64_t tile;
  {
    const int64_t per = 8 * (int64_t)a.ngroups, sg = blockIdx.x / per, in = blockIdx.x % per;
    const int64_t ntiles = (a.M + kRows6 - 1) / kRows6;
    if ((sg + 1) * 8 <= ntiles) { g = (int)(in / 8); tile = sg * 8 + in % 8; }
    else { const int64_t rest = blockIdx.x - sg * per; g = (int)(rest % a.ngroups); tile = sg * 8 + rest / a.ngroups; }
  }
  const int64_t row0 = tile * kRows6 + 32 * w;
  const int r = lane & 15, q = lane >> 4;
  const unsigned char *Wg = reinterpret_cast<const unsigned char *>(a.Wp) + (int64_t)g * (a.nslab + 2) * kSlabBytes;

  // LDS-DMA of one slab into ring buffer buf: kPieces pieces of 1 KiB, piece 8 i + w by wave w - the same count
  // for every wave, no branch.  (The prepped weights carry two zero slabs behind the last: the loop never asks.)
  auto stage = [&](int s, int buf) {
    const unsigned char *src = Wg + (int64_t)s * kSlabBytes + lane * 16;
    unsigned char *dst = smem6 + buf * kLdsSlab;
#pragma unroll
    for (int i = 0; i < kDmaPerWave; ++i) {
      const int piece = kWaves6 * i + w;
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void *)(src + piece * 1024),
          (__attribute__((address_space(3))) void *)(piece < 3 * NT ? dst + piece * 1024 : smem6 + kSink), 16, 0, 0);
    }
  };
  // slab s of this lane's two rows: out of A1 (whole slabs), the padded tail copy, or - past the end - the last
  // slab again (a wasted, in-range load: the loop body stays one basic block)
  const int64_t ra0 = row0 + r < a.M ? row0 + r : a.M - 1, ra1 = row0 + 16 + r < a.M ? row0 + 16 + r : a.M - 1;
  auto load_slab = [&](int s, float (&x0)[8], float (&x1)[8]) {
    const int sc = s < a.nslab ? s : a.nslab - 1;
    const bool tail = sc >= a.nfull;  // (wave-uniform: scalar selects)
    const bool copy = tail && a.Atail != nullptr;
    const float *base = copy ? a.Atail : a.A1;
    const int64_t ld = copy ? kSlabK : a.lda1;
    const int k0 = (copy ? 0 : (tail ? a.tail_k0 : kSlabK * sc)) + 8 * q;
    load_a8(base, ld, ra0, k0, x0);
    load_a8(base, ld, ra1, k0, x1);
  };

  f32x4 acc[2][NT];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#ifdef RM_NN6_STAMP
  if (tid == 0 && blockIdx.x < 4096) {
    rm_nn6_stamp_buf[4 * blockIdx.x + 0] = __builtin_amdgcn_s_memrealtime();
    rm_nn6_stamp_buf[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memtime();
  }
#endif

  // prologue: the A rows of slabs 0, 1 (register sets P, Q) and 2 (P again, once slab 0 is split) and the weights
  // of slabs 0 and 1 (ring buffers 0, 1) on their way
  float xp[2][8], xq[2][8];
  bf16x8 fr[2][6];  // the split A fragments of two slabs: [slab parity][h0 m0 l0 h1 m1 l1]
  load_slab(0, xp[0], xp[1]);
  stage(0, 0);
  load_slab(1, xq[0], xq[1]);
  if (kRing6 == 3) stage(1, 1);
  split8(xp[0], fr[0][0], fr[0][1], fr[0][2]);
  split8(xp[1], fr[0][3], fr[0][4], fr[0][5]);
  load_slab(2, xp[0], xp[1]);
  // weights of slab 0 landed (what was requested behind them may still be on its way)
  if (kRing6 == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kDmaPerWave + 4) : "memory");
  else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // One slab s (CUR = its parity): MFMAs on the fragments split during the PREVIOUS slab, and between them - the
  // matrix pipe holds the vector issue for only half of an MFMA's 16 cycles - the splitting of slab s + 1's rows
  // (loaded two slabs ago); then slab s + 3's rows into the registers just split and slab s + 2's weights into the
  // ring buffer read during slab s - 1.  (With the split ahead of the MFMAs all 8 waves of the block - in step
  // behind the barrier - did their ~130 vector instructions while the matrix pipe idled: 2,600 of 7,600 cycles
  // per slab.)
  auto slab = [&](int s, auto cur, float (&xn0)[8], float (&xn1)[8]) {
    constexpr int C = decltype(cur)::value, Nx = C ^ 1;
    if (!(RM_NN6_ABL & 16)) stage(s + kAhead, (s + kAhead) % kRing6);
    const unsigned char *ws = smem6 + (s % kRing6) * kLdsSlab + lane * 16;
    if (!(RM_NN6_ABL & 4)) {
      split8(xn0, fr[Nx][0], fr[Nx][1], fr[Nx][2]);
      split8(xn1, fr[Nx][3], fr[Nx][4], fr[Nx][5]);
    }
    if (!(RM_NN6_ABL & 2)) load_slab(s + 3, xn0, xn1);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int jj = (RM_NN6_ABL & 8) ? 0 : j;
      const bf16x8 bh = *reinterpret_cast<const bf16x8 *>(ws + (0 * NT + jj) * 1024);
      const bf16x8 bm = *reinterpret_cast<const bf16x8 *>(ws + (1 * NT + jj) * 1024);
      const bf16x8 bl = *reinterpret_cast<const bf16x8 *>(ws + (2 * NT + jj) * 1024);
      // the small products first, the leading one last.  The WEIGHT fragment is the MFMA's A operand: the result
      // tile is D[m = column 4 q + i of the tile][n = batch row r] - a lane ends up with four CONSECUTIVE output
      // columns of one batch row (one 16-byte store, bias / aux as one 16-byte load).  The two row tiles'
      // accumulators alternate: no MFMA reads the result of the one issued just before it.
      const bf16x8 ah0 = fr[C][0], am0 = fr[C][1], al0 = fr[C][2], ah1 = fr[C][3], am1 = fr[C][4], al1 = fr[C][5];
      f32x4 c0 = acc[0][j], c1 = acc[1][j];
      if (RM_NN6_ABL & 32) {
        c0[0] += (float)al0[0] + (float)bh[0] + (float)bm[1] + (float)bl[2] + (float)ah0[1] + (float)am0[2];
        c1[0] += (float)al1[0] + (float)bh[0] + (float)bm[1] + (float)bl[2] + (float)ah1[1] + (float)am1[2];
      } else {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, al0, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, al1, c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl, ah0, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl, ah1, c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bm, am0, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bm, am1, c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, am0, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, am1, c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bm, ah0, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bm, ah1, c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, ah0, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, ah1, c1, 0, 0, 0);
      }
      acc[0][j] = c0;
      acc[1][j] = c1;
    }
    // the order hipcc is to emit: after every MFMA one vector instruction of the next slab's split (an MFMA holds
    // the vector issue for half of its 16 cycles; left alone the scheduler puts the ~130 split instructions in
    // front of the 156 MFMAs, and the two waves of a SIMD - in step behind the barrier - then both leave the
    // matrix pipe idle), the three fragment reads of the next tile behind each tile's twelve
    if (!(RM_NN6_ABL & 64)) {
#pragma unroll
      for (int j = 0; j < NT; ++j) {
#pragma unroll
        for (int i = 0; i < 12; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
          __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);  // VALU
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);    // DS read
      }
    }
    // (a raw barrier: __syncthreads() carries a fence, i.e. s_waitcnt vmcnt(0) - it would wait for the rows and
    // weights just requested for the slabs ahead)
    if (RM_NN6_ABL & (2 | 16)) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(kVmPerSlab) : "memory");
    __builtin_amdgcn_s_barrier();
  };
  int s = 0;
  for (; s + 1 < a.nslab; s += 2) {
    slab(s, std::integral_constant<int, 0>{}, xq[0], xq[1]);      // (splits slab s + 1 = Q, refills Q with s + 3)
    slab(s + 1, std::integral_constant<int, 1>{}, xp[0], xp[1]);  // (splits slab s + 2 = P, refills P with s + 4)
  }
  if (s < a.nslab) slab(s, std::integral_constant<int, 0>{}, xq[0], xq[1]);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the wasted loads past the last slab)
#ifdef RM_NN6_STAMP
  if (tid == 0 && blockIdx.x < 4096) {
    rm_nn6_stamp_buf[4 * blockIdx.x + 2] = __builtin_amdgcn_s_memrealtime();
    rm_nn6_stamp_buf[4 * blockIdx.x + 3] = __builtin_amdgcn_s_memtime();
  }
#endif

  // epilogue: lane (r, q) holds columns 4 q .. 4 q + 3 of every tile for batch rows r (t = 0) and 16 + r (t = 1)
  const bool vec4 = (a.ldc & 3) == 0 && (a.N & 3) == 0 && (a.aux1 == nullptr || (a.ld1 & 3) == 0);
  float dot[2] = {0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int64_t row = row0 + 16 * t + r;
    if (row >= a.M) continue;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int col = 16 * (NT * g + j) + 4 * q;
      if (col >= a.N) continue;
      float v[4], b[4] = {0.f, 0.f, 0.f, 0.f}, x[4] = {0.f, 0.f, 0.f, 0.f};
      const bool whole = vec4 && col + 4 <= a.N;
      if (whole) {
        if (a.bias != nullptr) *reinterpret_cast<float4 *>(b) = *reinterpret_cast<const float4 *>(a.bias + col);
        if (a.aux1 != nullptr) *reinterpret_cast<float4 *>(x) = *reinterpret_cast<const float4 *>(a.aux1 + row * a.ld1 + col);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (col + i >= a.N) continue;
          if (a.bias != nullptr) b[i] = a.bias[col + i];
          if (a.aux1 != nullptr) x[i] = a.aux1[row * a.ld1 + col + i];
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float u = acc[t][j][i] + b[i];
        if (a.epi == RM_DENSE_BIAS_ACT) u = act6(u, a.act);
        else if (a.epi == RM_DENSE_MUL_ACTGRAD) u *= actgrad6(x[i], a.act);
        else u += x[i];
        v[i] = u;
      }
      if (a.dot_w != nullptr) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (col + i < a.N) dot[t] += v[i] * a.dot_w[col + i];
      }
      if ((RM_NN6_ABL & 1) && v[0] != 12345.678f) continue;
      if (whole) {
        *reinterpret_cast<float4 *>(a.C + row * a.ldc + col) = *reinterpret_cast<const float4 *>(v);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (col + i < a.N) a.C[row * a.ldc + col + i] = v[i];
      }
    }
  }
  if (a.dot_w != nullptr) {
    // the row's dot product over this group's columns: the four column quarters (q) of a row sit 16 lanes apart
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float d = dot[t];
      d += __shfl_xor(d, 16, 64);
      d += __shfl_xor(d, 32, 64);
      const int64_t row = row0 + 16 * t + r;
      if (q == 0 && row < a.M) a.dot_part[(int64_t)g * a.M + row] = d;
    }
  }
}

// out[b] = w0 + the groups' partial dot products in group order
__global__ void dense6_dot_finish_kernel(const float *__restrict__ part, int ngroups, int64_t M,
                                         const float *__restrict__ w0, float *__restrict__ out) {
  for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < M; b += (int64_t)gridDim.x * blockDim.x) {
    float v = w0 != nullptr ? w0[0] : 0.f;
    for (int g = 0; g < ngroups; ++g) v += part[(int64_t)g * M + b];
    out[b] = v;
  }
}

}  // namespace

#ifdef RM_NN6_STAMP
extern "C" int rm_debug_nn6_stamps(unsigned long long *host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(rm_nn6_stamp_buf), sizeof(unsigned long long) * n);
}
#endif

static inline int64_t dense6_w_floats(int K, int N) {
  const int nslab = (K + kSlabK - 1) / kSlabK;
  const int nt = (N + 15) / 16;
  const int ngroups = (nt + kMaxNT6 - 1) / kMaxNT6;
  return ((int64_t)ngroups * (nslab + 2) * ((3 * kMaxNT6 + 7) / 8 * 8) * 1024 + 3) / 4;
}

// floats of workspace: the split weights + the padded copy of the ragged end of the activations [M, 32] (a second
// piece A2) + the per-group partial row dots
extern "C" int64_t rm_dense6_workspace(int K, int N, int64_t M) {
  if (K <= 0 || N <= 0 || M < 0) return 0;
  const int ngroups = ((N + 15) / 16 + kMaxNT6 - 1) / kMaxNT6;
  return dense6_w_floats(K, N) + M * kSlabK + (int64_t)ngroups * M + 64;
}

extern "C" int rm_dense_fwd6(const float *A1, int64_t lda1, int K1, const float *A2, int64_t lda2, int K2,
                             const float *W, int64_t ldw, int w_transposed, int N, const float *bias, int epilogue,
                             int act, const float *aux1, int64_t ld_aux1, int64_t M, float *C, int64_t ldc,
                             const float *dot_w, const float *dot_w0, float *dot_out, float *workspace,
                             rm_stream_t stream) {
  RM_REQUIRE(M >= 0 && K1 > 0 && K2 >= 0 && N > 0, "rm_dense_fwd6: bad sizes");
  if (M == 0) return RM_OK;
  RM_REQUIRE(A1 && W && C && workspace && rm_aligned16(workspace), "rm_dense_fwd6: NULL / unaligned argument");
  RM_REQUIRE(K2 == 0 || A2, "rm_dense_fwd6: a second piece needs A2");
  RM_REQUIRE(epilogue >= RM_DENSE_BIAS_ACT && epilogue <= RM_DENSE_ADD, "rm_dense_fwd6: epilogue not covered");
  RM_REQUIRE(epilogue != RM_DENSE_MUL_ACTGRAD || aux1, "rm_dense_fwd6: MUL_ACTGRAD needs aux1");
  RM_REQUIRE(rm_aligned16(A1) && lda1 % 4 == 0, "rm_dense_fwd6: A1 rows must be 16-byte aligned");
  RM_REQUIRE(!dot_w == !dot_out, "rm_dense_fwd6: dot_w and dot_out come together");
  const int K = K1 + K2;
  const int nslab = (K + kSlabK - 1) / kSlabK;
  const int nfull = K1 / kSlabK;  // whole slabs of A1
  if (nslab - nfull > 1) {
    rm_set_error("rm_dense_fwd6: the ragged end (K1 %% 32 columns + K2) must fit one slab of 32");
    return RM_EUNSUPPORTED;
  }
  // the ragged end: one piece with at least a slab of 4-aligned columns -> the last slab re-reads [K - 32, K) in
  // place (its weights zeroed where the slab before has been); otherwise a padded copy
  const bool ragged = nslab > nfull;
  const bool overlap = ragged && K2 == 0 && K1 >= kSlabK && K1 % 4 == 0;
  const int nt = (N + 15) / 16;
  const int ngroups = (nt + kMaxNT6 - 1) / kMaxNT6;
  hipStream_t st = (hipStream_t)stream;
  __bf16 *Wp = reinterpret_cast<__bf16 *>(workspace);
  float *Atail = workspace + (dense6_w_floats(K, N) + 3) / 4 * 4;
  float *dot_part = Atail + M * kSlabK;
  hipLaunchKernelGGL((dense6_prep_kernel<kMaxNT6>),
                     dim3(rm_grid_cap(((int64_t)ngroups * nslab * kMaxNT6 * 512 + 255) / 256, 2048)), dim3(256), 0,
                     st, W, ldw, w_transposed, K, N, nslab, ngroups, overlap ? K - kSlabK : -1, Wp);
  if (ragged && !overlap)
    hipLaunchKernelGGL(dense6_tail_kernel, dim3(rm_grid_cap((M * kSlabK + 255) / 256, 256 * 8)), dim3(256), 0, st, A1,
                       lda1, K1, A2, lda2, K2, nfull * kSlabK, M, Atail);
  // (starting the second resident of every CU 4 .. 30 us late changed nothing: 351 .. 367 us against 355)
  NN6Args a{A1, (ragged && !overlap) ? Atail : nullptr, lda1, nfull, overlap ? K - kSlabK : 0, dot_w,
            dot_part, Wp, N, nslab, ngroups, bias, epilogue, act, aux1, ld_aux1, M, C, ldc};
  const int64_t ntiles = (M + kRows6 - 1) / kRows6;
  const size_t smem = (size_t)kRing6 * 3 * kMaxNT6 * 1024 + 1024;
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(dense_nn6_kernel<kMaxNT6>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL((dense_nn6_kernel<kMaxNT6>), dim3((unsigned)(ntiles * ngroups)), dim3(kThreads6), smem, st, a);
  if (dot_w)
    hipLaunchKernelGGL(dense6_dot_finish_kernel, dim3(rm_grid_cap((M + 255) / 256, 1024)), dim3(256), 0, st, dot_part,
                       ngroups, M, dot_w0, dot_out);
  RM_CHECK_LAUNCH("rm_dense_fwd6");
  return RM_OK;
}

// ---- "TN": dW[K, N] (+)= [A1 | A2]^T . G, reduction over the batch, db[N] = column sums of G ------------------
// Replaces the weight-gradient GEMM of DNN's backward (tf.gradients of layers.py:594-602) like rm_dense_wgrad's f32
// path.  Both operands are activations: both are split on the fly.  An MFMA operand fragment here is 8 CONSECUTIVE
// BATCH ROWS of one column - strided in the row-major activations - so a slab (32 batch rows) goes through LDS as
// transposed bf16 planes [piece][column][32 batch]: a thread loads one column's 8 rows (8 dword loads, coalesced
// across the wave's 64 columns), splits them and writes three 16-byte pieces; a fragment read is one ds_read_b128
// (16 columns x 64 bytes: a linear KiB, conflict-free).  The split is done ONCE per block and slab and shared by its
// 8 waves.  Block = 224 K-columns (14 tiles) x 208 N-columns (13 tiles) x a range of slabs; wave (wk, wn) owns 7
// K-tiles x the N-tiles wn, wn + 4, wn + 8 (, 12): the four G fragment triples stay in registers, the A triples
// stream.  G is the MFMA's A operand: a lane ends with 4 consecutive N of one K row (16-byte stores).  The blocks'
// partial tiles go to the workspace, dense6_tn_reduce_kernel adds them in split order (deterministic).
namespace {

#ifndef RM_TN6_ABL
#define RM_TN6_ABL 0  // ablation builds (WRONG results): 1 no MFMAs, 2 no loads in the loop, 4 no split / LDS writes
#endif
constexpr int kTnKT = 14, kTnNT = 13;
constexpr int kTnKC = 16 * kTnKT, kTnNC = 16 * kTnNT;           // 224, 208
constexpr int kTnUnits = 4 * (kTnKC + kTnNC);                   // (column, 8-row group) units per slab: 1728
constexpr int kTnPerThread = (kTnUnits + 511) / 512;            // 4 (the last round only for 192 threads)
constexpr int kTnPlaneA = kTnKC * 64, kTnPlaneG = kTnNC * 64;   // bytes of one bf16 plane
constexpr int kTnLds = 3 * (kTnPlaneA + kTnPlaneG);             // 82,944

struct TN6Args {
  const float *A1, *A2, *G, *G2;  // G2 [M, N2]: a second piece of gradient columns behind G's N (or NULL)
  int64_t lda1, lda2, ldg, ldg2;
  int K1, K2, N, N2;
  int64_t M;
  int nkh, nnh, nsplit;
  int64_t slabs_per_split, nslab;
  float *part;     // [nsplit][nkh][nnh][224][208]
  float *db_part;  // [nsplit][nnh][208] or NULL
};

template <bool RAGGED>
__global__ __launch_bounds__(512) void dense_tn6_kernel(TN6Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem6[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  // blockIdx -> (output tile, batch split): the nkh x nnh tiles of ONE split read the same rows of A and G; they
  // are put 8 blocks apart = on the same XCD (round-robin placement), so those rows come from HBM about once
  int tile_id, sp;
  {
    const int nt = a.nkh * a.nnh, per = 8 * nt, sg = blockIdx.x / per, in = blockIdx.x % per;
    if ((sg + 1) * 8 <= a.nsplit) { tile_id = in / 8; sp = sg * 8 + in % 8; }
    else { const int rest = blockIdx.x - sg * per; tile_id = rest % nt; sp = sg * 8 + rest / nt; }
  }
  const int nh = tile_id % a.nnh, kh = tile_id / a.nnh;
  const int K = a.K1 + a.K2;
  const int kcol0 = kTnKC * kh, ncol0 = kTnNC * nh;

  // this thread's units: (column, 8-row group) of the A half or the G half - the same ones in every slab
  const float *uptr[kTnPerThread];
  int64_t uld[kTnPerThread];
  float ukeep[kTnPerThread];
  int ulds[kTnPerThread], urow[kTnPerThread];
  bool uisg[kTnPerThread];
#pragma unroll
  for (int i = 0; i < kTnPerThread; ++i) {
    int u = tid + 512 * i;
    u = u < kTnUnits ? u : kTnUnits - 1;  // (the spare threads of the last round repeat the last unit)
    const bool isg = u >= 4 * kTnKC;
    const int v = isg ? u - 4 * kTnKC : u;
    const int cols = isg ? kTnNC : kTnKC;
    const int col = v % cols, g = v / cols;
    const int c = (isg ? ncol0 : kcol0) + col;
    const float *p;
    int64_t ld;
    bool live;
    if (isg && c < a.N) { live = true; p = a.G + c; ld = a.ldg; }
    else if (isg && c < a.N + a.N2) { live = true; p = a.G2 + (c - a.N); ld = a.ldg2; }
    else if (isg) { live = false; p = a.G; ld = a.ldg; }
    else if (c < a.K1) { live = true; p = a.A1 + c; ld = a.lda1; }
    else if (c < K) { live = true; p = a.A2 + (c - a.K1); ld = a.lda2; }
    else { live = false; p = a.A1; ld = a.lda1; }
    uptr[i] = p; uld[i] = ld; ukeep[i] = live ? 1.f : 0.f; urow[i] = 8 * g; uisg[i] = isg;
    // a column's four 16-byte pieces (8-row groups) are rotated by col / 4: the 64 columns a wave writes in one
    // instruction (same g) then spread over all 64 banks (unrotated: 16 banks, 4x the cycles); a fragment read
    // still covers its 16 columns x 64 bytes as one contiguous KiB
    ulds[i] = (isg ? 3 * kTnPlaneA : 0) + col * 64 + ((g + (col >> 2)) & 3) * 16;
  }
  auto load_units = [&](int64_t slab, float (&x)[kTnPerThread][8]) {
    const int64_t b0 = slab * 32;
#pragma unroll
    for (int i = 0; i < kTnPerThread; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int64_t row = b0 + urow[i] + e;
        x[i][e] = uptr[i][(RAGGED && row >= a.M ? a.M - 1 : row) * uld[i]];
      }
  };

  // wave (wk, wn): K-tiles 7 wk .. 7 wk + 6 x N-tiles wn, wn + 4, wn + 8; the 13th N-tile's 14 tiles are dealt out:
  // K-tile 7 wk + t goes to the wave with wn == t % 4 (23 / 23 / 23 / 22 tiles per wave; wn + 12 for wn = 0 only
  // was 28 / 21 / 21 / 21 - the SIMD with the two big waves set the pace of every slab)
  const int wk = w >> 2, wn = w & 3;
  f32x4 acc[7][3], acx[2];
#pragma unroll
  for (int t = 0; t < 7; ++t)
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  acx[0] = acx[1] = f32x4{0.f, 0.f, 0.f, 0.f};
  float dbacc[kTnPerThread] = {0.f, 0.f, 0.f, 0.f};
  auto frag = [&](int base, int plane, int p, int tile) {
    const int col = 16 * tile + r;
    return *reinterpret_cast<const bf16x8 *>(smem6 + base + p * plane + col * 64 + ((q + (col >> 2)) & 3) * 16);
  };
  auto mma6 = [&](f32x4 c, const bf16x8 (&gq)[3], const bf16x8 (&aq)[3]) {
    if (RM_TN6_ABL & 1) {
      c[0] += (float)gq[0][0] + (float)gq[1][1] + (float)gq[2][2] + (float)aq[0][3] + (float)aq[1][4] + (float)aq[2][5];
      return c;
    }
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gq[0], aq[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gq[2], aq[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gq[1], aq[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gq[0], aq[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gq[1], aq[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gq[0], aq[0], c, 0, 0, 0);
    return c;
  };

  const int64_t s0 = (int64_t)sp * a.slabs_per_split;
  const int64_t s1 = s0 + a.slabs_per_split < a.nslab ? s0 + a.slabs_per_split : a.nslab;
  float x[kTnPerThread][8];
  if (s0 < s1) load_units(s0, x);
  for (int64_t s = s0; s < s1; ++s) {
    // split this slab's values (loaded during the previous slab's MFMAs) into the LDS planes
    const int64_t b0 = s * 32;
    __builtin_amdgcn_s_barrier();  // (every wave is done reading the previous slab's planes)
#pragma unroll
    for (int i = 0; i < kTnPerThread; ++i) {
      if ((RM_TN6_ABL & 4) && s != s0) continue;
      float y[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) y[e] = (!RAGGED || b0 + urow[i] + e < a.M) ? x[i][e] * ukeep[i] : 0.f;
      if (uisg[i] && a.db_part != nullptr && (i < kTnPerThread - 1 || tid + 512 * i < kTnUnits))
        dbacc[i] += ((y[0] + y[1]) + (y[2] + y[3])) + ((y[4] + y[5]) + (y[6] + y[7]));
      bf16x8 h, m, l;
      split8(y, h, m, l);
      unsigned char *d = smem6 + ulds[i];
      const int plane = uisg[i] ? kTnPlaneG : kTnPlaneA;
      *reinterpret_cast<bf16x8 *>(d) = h;
      *reinterpret_cast<bf16x8 *>(d + plane) = m;
      *reinterpret_cast<bf16x8 *>(d + 2 * plane) = l;
    }
    // the next slab's values, on their way during the MFMAs below (unconditional - the last slab is requested
    // again - so that the loads stay in this basic block: hipcc sinks a load under an `if` to its use)
    if (!(RM_TN6_ABL & 2)) load_units(s + 1 < s1 ? s + 1 : s, x);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // G fragment triples of this wave's N-tiles (and of the shared 13th)
    bf16x8 gf[4][3];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int p = 0; p < 3; ++p) gf[j][p] = frag(3 * kTnPlaneA, kTnPlaneG, p, j < 3 ? wn + 4 * j : 12);
#pragma unroll
    for (int t = 0; t < 7; ++t) {
      bf16x8 af[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) af[p] = frag(0, kTnPlaneA, p, 7 * wk + t);
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[t][j] = mma6(acc[t][j], gf[j], af);
      if ((t & 3) == wn) acx[t >> 2] = mma6(acx[t >> 2], gf[3], af);  // (wave-uniform)
    }
  }

  // the block's partial tile: lane (r, q) holds N-columns 4 q .. 4 q + 3 of K-row r of every tile
  float *pt = a.part + (((int64_t)sp * a.nkh + kh) * a.nnh + nh) * (int64_t)(kTnKC * kTnNC);
#pragma unroll
  for (int t = 0; t < 7; ++t) {
    const int kt = 7 * wk + t;
#pragma unroll
    for (int j = 0; j < 3; ++j)
      *reinterpret_cast<f32x4 *>(pt + (16 * kt + r) * kTnNC + 16 * (wn + 4 * j) + 4 * q) = acc[t][j];
    if ((t & 3) == wn) *reinterpret_cast<f32x4 *>(pt + (16 * kt + r) * kTnNC + 16 * 12 + 4 * q) = acx[t >> 2];
  }
  if (a.db_part != nullptr && kh == 0) {
    // column sums of G: the four 8-row groups of a column sit in four threads' accumulators -> through LDS
    __builtin_amdgcn_s_barrier();
    float *red = reinterpret_cast<float *>(smem6);  // [4][208]
#pragma unroll
    for (int i = 0; i < kTnPerThread; ++i) {
      const int u = tid + 512 * i;
      if (u >= 4 * kTnKC && u < kTnUnits) red[u - 4 * kTnKC] = dbacc[i];  // index = g * 208 + col
    }
    __syncthreads();
    if (tid < kTnNC)
      a.db_part[((int64_t)sp * a.nnh + nh) * kTnNC + tid] =
          (red[tid] + red[kTnNC + tid]) + (red[2 * kTnNC + tid] + red[3 * kTnNC + tid]);
  }
}

// dW[k][n] (+)= sum over the splits (in order) of the partial tiles; db[n] likewise; the columns behind N (a second
// piece of gradient columns) go to dW2[k][n - N]
__global__ void dense6_tn_reduce_kernel(const float *__restrict__ part, const float *__restrict__ db_part, int K,
                                        int N, int N2, int nkh, int nnh, int nsplit, float *__restrict__ dW,
                                        int64_t lddw, float *__restrict__ dW2, int64_t lddw2, int accumulate,
                                        float *__restrict__ db) {
  const int Nt = N + N2;
  const int64_t total = (int64_t)K * Nt;
  const int64_t tile = (int64_t)kTnKC * kTnNC;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total + N;
       t += (int64_t)gridDim.x * blockDim.x) {
    if (t < total) {
      const int k = (int)(t / Nt), n = (int)(t % Nt);
      const int kh = k / kTnKC, nh = n / kTnNC;
      const float *p = part + ((int64_t)kh * nnh + nh) * tile + (int64_t)(k % kTnKC) * kTnNC + n % kTnNC;
      float v = 0.f;
      for (int s = 0; s < nsplit; ++s) v += p[(int64_t)s * nkh * nnh * tile];
      float *d = n < N ? dW + (int64_t)k * lddw + n : dW2 + (int64_t)k * lddw2 + (n - N);
      *d = accumulate ? *d + v : v;
    } else if (db != nullptr) {
      const int n = (int)(t - total);
      const float *p = db_part + (int64_t)(n / kTnNC) * kTnNC + n % kTnNC;
      float v = 0.f;
      for (int s = 0; s < nsplit; ++s) v += p[(int64_t)s * nnh * kTnNC];
      db[n] = v;
    }
  }
}

struct Tn6Plan {
  int nkh, nnh, nsplit;
  int64_t nslab, per;
};
Tn6Plan tn6_plan(int K, int N, int64_t M) {
  Tn6Plan p;
  p.nkh = (K + kTnKC - 1) / kTnKC;
  p.nnh = (N + kTnNC - 1) / kTnNC;
  p.nslab = (M + 31) / 32;
  int64_t want = 256 / (p.nkh * p.nnh);  // one block per CU: every split costs a partial tile written and read again
  if (want < 1) want = 1;
  if (want > p.nslab) want = p.nslab > 0 ? p.nslab : 1;
  p.per = (p.nslab + want - 1) / want;
  if (p.per < 1) p.per = 1;
  p.nsplit = (int)((p.nslab + p.per - 1) / p.per);
  if (p.nsplit < 1) p.nsplit = 1;
  return p;
}

}  // namespace

extern "C" int64_t rm_dense_wgrad6_workspace(int K, int N, int64_t M) {
  if (K <= 0 || N <= 0 || M <= 0) return 0;
  const Tn6Plan p = tn6_plan(K, N, M);
  return (int64_t)p.nsplit * p.nkh * p.nnh * kTnKC * kTnNC + (int64_t)p.nsplit * p.nnh * kTnNC + 64;
}

extern "C" int rm_dense_wgrad6(const float *A1, int64_t lda1, int K1, const float *A2, int64_t lda2, int K2,
                               const float *G, int64_t ldg, int N, const float *G2, int64_t ldg2, int N2, int64_t M,
                               float *dW, int64_t lddw, float *dW2, int64_t lddw2, int accumulate, float *db,
                               float *workspace, int64_t workspace_floats, rm_stream_t stream) {
  RM_REQUIRE(M >= 0 && K1 > 0 && K2 >= 0 && N > 0 && N2 >= 0, "rm_dense_wgrad6: bad sizes");
  RM_REQUIRE(A1 && G && dW && workspace && rm_aligned16(workspace), "rm_dense_wgrad6: NULL / unaligned argument");
  RM_REQUIRE(K2 == 0 || A2, "rm_dense_wgrad6: K2 > 0 needs A2");
  RM_REQUIRE(N2 == 0 || (G2 && dW2 && ldg2 >= N2 && lddw2 >= N2), "rm_dense_wgrad6: N2 > 0 needs G2 and dW2");
  RM_REQUIRE(lda1 >= K1 && (K2 == 0 || lda2 >= K2) && ldg >= N && lddw >= N,
             "rm_dense_wgrad6: a leading dimension is too small");
  const int K = K1 + K2, Nt = N + N2;
  hipStream_t st = (hipStream_t)stream;
  if (M == 0) {
    if (!accumulate) {
      (void)hipMemset2DAsync(dW, lddw * 4, 0, (size_t)N * 4, K, st);
      if (N2) (void)hipMemset2DAsync(dW2, lddw2 * 4, 0, (size_t)N2 * 4, K, st);
    }
    if (db) (void)hipMemsetAsync(db, 0, (size_t)N * 4, st);
    return RM_OK;
  }
  RM_REQUIRE(workspace_floats >= rm_dense_wgrad6_workspace(K, Nt, M), "rm_dense_wgrad6: workspace too small");
  const Tn6Plan p = tn6_plan(K, Nt, M);
  float *part = workspace;
  float *db_part = db ? workspace + (int64_t)p.nsplit * p.nkh * p.nnh * kTnKC * kTnNC : nullptr;
  TN6Args a{A1, A2, G, G2, lda1, lda2, ldg, ldg2, K1, K2, N, N2, M, p.nkh, p.nnh, p.nsplit, p.per, p.nslab, part,
            db_part};
  const dim3 grid((unsigned)(p.nsplit * p.nkh * p.nnh));
  if (M % 32 != 0) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(dense_tn6_kernel<true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, kTnLds);
    hipLaunchKernelGGL(dense_tn6_kernel<true>, grid, dim3(512), kTnLds, st, a);
  } else {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(dense_tn6_kernel<false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, kTnLds);
    hipLaunchKernelGGL(dense_tn6_kernel<false>, grid, dim3(512), kTnLds, st, a);
  }
  hipLaunchKernelGGL(dense6_tn_reduce_kernel, dim3(rm_grid_cap(((int64_t)K * Nt + N + 255) / 256, 2048)), dim3(256), 0,
                     st, part, db_part, K, N, N2, p.nkh, p.nnh, p.nsplit, dW, lddw, dW2, lddw2, accumulate, db);
  RM_CHECK_LAUNCH("rm_dense_wgrad6");
  return RM_OK;
}
