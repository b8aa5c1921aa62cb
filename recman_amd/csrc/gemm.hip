// Wide dense layers on the f32 MFMA (v_mfma_f32_32x32x2_f32) - the GEMMs of a DNN whose hidden
// widths exceed the fused skinny-MLP kernel (mlp.hip): DCN's deep_hidden_units (400, 400), and
// the matrix form of the cross layer  x_{l+1} = x0 o (W x_l + b) + x_l.
// Replaces tf.matmul + bias + activation of DNN.__call__ (recman/tf/core/layers.py:594-602)
// and the three GEMMs of its gradient.  hipBLASLt ran these shapes at 46-58 % of the f32 MFMA
// peak and needed extra launches for the K=13 dense-input piece, the bias/activation and the
// activation gradient; here x = [A1 | A2] is read in place and the epilogues are fused.
//
//   rm_dense_fwd   C[M,N]  = epilogue([A1 | A2][M,K] . op(W))      "NN": M = batch
//   rm_dense_wgrad dW[K,N] = [A1 | A2]^T . G[M,N]                   "TN": reduction over batch
//
// NN kernel: block = 4 waves (one per SIMD) = 128 batch rows x ONE column group of <= 7 32-column blocks
// (a 13-block N = 400 is two groups, 7 + 6), two blocks per CU: the blocks drift out of phase, so one's
// MFMAs cover the other's barriers, staging and epilogue.  The A tile [128][32 k] and the group's weight
// chunk [16 k][<= 224 cols] are staged in LDS, double-buffered; the weights are pre-arranged by
// dense_prep_kernel so that a lane's NT operands are two ds_read_b128.
// TN kernel: no LDS at all - both operands are row-contiguous along the MFMA's M / N index, so
// every wave loads them coalesced (128 B per half-wave) in operand layout; the batch is split
// into slabs, partial tiles go to a workspace and a second kernel reduces them (deterministic).
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "rm_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

#ifndef RM_GEMM_KC
#define RM_GEMM_KC 16
#endif
#ifndef RM_GEMM_SGB
#define RM_GEMM_SGB 0
#endif
// ablation builds of the NN kernel (WRONG results; tools/bench_dense.py + `python -m recman_amd.build --out`):
// 1 no epilogue stores, 2 no per-chunk barrier, 4 no weight loads, 8 no A loads / commits,
// 16 no B operand reads in the k-steps, 32 no weight ds_writes
#ifndef RM_NN_ABL
#define RM_NN_ABL 0
#endif
// diagnostic build (-DRM_NN_STAMP): wave 0 of every NN block records s_memrealtime (100 MHz) at block
// start / after the prologue / after the chunk loop / after the epilogue + its HW_ID; read back with
// rm_debug_nn_stamps (tools/probe/nn_stamps.py).  Never defined in the product build.
#ifdef RM_NN_STAMP
__device__ unsigned long long rm_nn_stamp_buf[8 * 8192];
#define RM_STAMP(i_)                                                                              \
  if (tid == 0 && blockIdx.x < 8192) rm_nn_stamp_buf[8 * blockIdx.x + (i_)] = __builtin_amdgcn_s_memrealtime()
#else
#define RM_STAMP(i_)
#endif
constexpr int KC = RM_GEMM_KC;               // k per staged weight chunk
constexpr int kRowsPerBlock = 128;           // TN kernel: 4 row groups x 32 rows of K
constexpr int kNNRows = 128;                 // NN kernel: 4 row groups x 32 batch rows per block
constexpr int kNNThreads = 256;              // 4 waves = one per SIMD; two blocks per CU
constexpr int kMaxNB = 14;                   // 32-column blocks per column tile (2 x 7)
constexpr int kTileCols = kMaxNB * 32;       // 448
constexpr int WCHG = KC * 2 * 32 * 4;        // floats per prepped chunk of ONE column group: [k][half][c][4] = 16 KB
constexpr int WCH = 2 * WCHG;                // both groups of a column tile
constexpr int kThreads = 512;

__device__ __forceinline__ float act_apply(float v, int act) {
  if (act == RM_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == RM_ACT_LEAKY_RELU) return v > 0.f ? v : 0.2f * v;
  return v;
}
__device__ __forceinline__ float act_grad_from_out(float o, int act) {
  if (act == RM_ACT_RELU) return o > 0.f ? 1.f : 0.f;
  if (act == RM_ACT_LEAKY_RELU) return o > 0.f ? 1.f : 0.2f;
  return 1.f;
}

struct ColTile {
  int nb, nt0;  // 32-col blocks in the tile; blocks of column group 0 (group 1 has nb - nt0)
};
__host__ __device__ inline ColTile col_tile(int N, int ct) {
  const int nbt = (N + 31) / 32;
  int nb = nbt - ct * kMaxNB;
  nb = nb < kMaxNB ? nb : kMaxNB;
  return ColTile{nb, (nb + 1) / 2};
}

// Wp[ct][g][chunk][k][half][c][j] = op(W)[chunk*16 + k][col(ct, g, nt = half*4 + j, c)], zero padded:
// the chunks of ONE column group are contiguous (a block stages only its own group's 16 KB per chunk)
__global__ void dense_prep_kernel(const float *__restrict__ W, int64_t ldw, int trans, int K, int N,
                                  int nch, int nct, float *__restrict__ Wp) {
  const int64_t total = (int64_t)nct * 2 * nch * WCHG;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int j = t & 3, c = (t >> 2) & 31, half = (t >> 7) & 1, k = (t >> 8) & (KC - 1);
    const int64_t chunk = t / WCHG;
    const int ch = (int)(chunk % nch), g = (int)((chunk / nch) & 1), ct = (int)(chunk / (2 * nch));
    const ColTile tl = col_tile(N, ct);
    const int nt = half * 4 + j;
    const int ntw = g == 0 ? tl.nt0 : tl.nb - tl.nt0;
    const int col = ct * kTileCols + 32 * (g == 0 ? nt : tl.nt0 + nt) + c;
    const int kk = ch * KC + k;
    float v = 0.f;
    if (nt < ntw && kk < K && col < N) v = trans ? W[(int64_t)col * ldw + kk] : W[(int64_t)kk * ldw + col];
    Wp[t] = v;
  }
}

struct NNArgs {
  const float *A1, *A2;
  int64_t lda1, lda2;
  int K1, K2;
  int a_vec;  // A1 rows are 16-byte aligned (float4 loads); otherwise every chunk loads per element
  const float *Wp;
  int N, nch;
  const float *bias;
  int epi, act;
  const float *aux1, *aux2;
  int64_t ld1, ld2;
  int64_t M;
  float *C, *C2;
  int64_t ldc, ldc2;
  int span, groups;  // blockIdx.x -> (row tile, column group): see dense_nn_kernel
  int stagger_ticks; // 100 MHz ticks the odd-group blocks of the first round wait before they start
};

constexpr int AQ = KC / 8;  // float4 of A fragments per lane per weight chunk

// One wave of the NN kernel with EXACTLY NT accumulators (no conditional MFMAs: a per-MFMA
// `if (nt < ntw)` cost 16 branches per chunk and fenced the scheduler).  ntw <= NT is the number
// of column blocks that really exist (the prepped weights of the others are zero).
template <int NT>
__device__ __forceinline__ void nn_wave(const NNArgs &a, float *Ws, int tid, int c, int h, int rg,
                                        int cg, int ct, const ColTile tl, int ntw, int64_t tile) {
  const int64_t row0 = tile * kNNRows + 32 * rg;
  const float *Wp = a.Wp + (int64_t)(ct * 2 + cg) * a.nch * WCHG;

  RM_STAMP(0);
#ifdef RM_NN_STAMP
  if (tid == 0 && blockIdx.x < 8192) {
    rm_nn_stamp_buf[8 * blockIdx.x + 4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_REG_HW_ID
    rm_nn_stamp_buf[8 * blockIdx.x + 6] = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
    rm_nn_stamp_buf[8 * blockIdx.x + 5] = cg;
  }
#endif
  // the accumulators START at bias[col] (a lane's 16 outputs of a column block share one column): no
  // bias loads or adds in the epilogue, where every VALU instruction is expensive (see below)
  f32x16 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int col = ct * kTileCols + 32 * (cg == 0 ? nt : tl.nt0 + nt) + c;
    const float bn = (a.bias != nullptr && nt < ntw) ? a.bias[col < a.N ? col : a.N - 1] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nt][r] = bn;
  }

  // A goes through LDS in FULL 128-byte lines: per group of 32 k the block stages its 128 rows once
  // (8 lanes per row) and the waves read their fragments from there.
  constexpr int AG = 32;                      // k per A group = 2 weight chunks
  constexpr int ALD = AG + 4;                 // LDS row stride of the A tile (floats)
  float *As = Ws + 2 * WCHG;                  // [2][kNNRows][ALD]
  const int64_t brow0 = tile * kNNRows;
  const int ngroups = (a.nch + 1) / 2;
  const int gfast = a.a_vec ? a.K1 / AG : 0;  // groups inside A1 that float4 loads can take
  const float *A2 = a.K2 > 0 ? a.A2 : a.A1;
  const int64_t lda2 = a.K2 > 0 ? a.lda2 : a.lda1;
  // this thread's four float4 of a group: f = tid + 256*i -> row f >> 3, piece f & 7.
  // Two loaders, picked at COMPILE time per loop (a run-time `if (g < gfast)` inside one loop made
  // hipcc join the two paths behind s_waitcnt vmcnt(0): every A group then waited for the weight
  // prefetch issued just before it AND for its own first load, profiles/r02_dense_gemm.md):
  //   fast: the group lies inside A1 with 16-byte rows -> one float4;
  //   slow: the ragged groups (x = [A1 | A2] boundary, end of K): both candidates of every element
  //   through raw buffer loads (unconditional by construction; rows past M read as 0) and the
  //   choice A1 / A2 / zero is made at commit time, behind the chunk's MFMAs.
  const int piece = tid & 7, arow = tid >> 3;  // + 32 i
  auto load_fast = [&](int g, int i) -> float4 {
    int64_t gr = brow0 + arow + 32 * i;
    gr = gr < a.M ? gr : a.M - 1;
    return *reinterpret_cast<const float4 *>(a.A1 + gr * a.lda1 + g * AG + 4 * piece);
  };
  const int64_t rows_here = a.M - brow0 < kNNRows ? a.M - brow0 : kNNRows;
  const rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.A1 + brow0 * a.lda1), 0,
                                                       (int)((rows_here - 1) * a.lda1 + a.K1) * 4, 0x00020000);
  const rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(A2 + brow0 * lda2), 0,
                                                       (int)((rows_here - 1) * lda2 + (a.K2 > 0 ? a.K2 : a.K1)) * 4,
                                                       0x00020000);
  auto load_slow = [&](int g, int i, float4 &x1, float4 &x2) {
    const int r = arow + 32 * i, k = g * AG + 4 * piece;
    float t1[4], t2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int kk = k + e, k2 = kk - a.K1;
      const int o1 = (r * (int)a.lda1 + (kk < a.K1 ? kk : 0)) * 4;
      const int o2 = (r * (int)lda2 + ((k2 >= 0 && k2 < a.K2) ? k2 : 0)) * 4;
      t1[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs1, o1, 0, 0));
      t2[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs2, o2, 0, 0));
    }
    x1 = make_float4(t1[0], t1[1], t1[2], t1[3]);
    x2 = make_float4(t2[0], t2[1], t2[2], t2[3]);
  };
  auto pick_slow = [&](int g, const float4 &x1, const float4 &x2) -> float4 {
    const int k = g * AG + 4 * piece;
    const float v1[4] = {x1.x, x1.y, x1.z, x1.w}, v2[4] = {x2.x, x2.y, x2.z, x2.w};
    float t[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int kk = k + e, k2 = kk - a.K1;
      t[e] = kk < a.K1 ? v1[e] : ((k2 < a.K2) ? v2[e] : 0.f);
    }
    return make_float4(t[0], t[1], t[2], t[3]);
  };
  const int a_lds = arow * ALD + 4 * piece;  // float4 slot i: + 32 i rows

  // prologue: weight chunk 0 and A group 0 to LDS
  constexpr int WQ = WCHG / 4 / kNNThreads;  // float4 of a weight chunk per thread
  static_assert(WQ == 4 && kNNRows * 8 / kNNThreads == 4, "the staging below is written out for 4 + 4 float4 per thread");
#pragma unroll
  for (int q = 0; q < WQ; ++q)
    *reinterpret_cast<float4 *>(Ws + (tid + q * kNNThreads) * 4) =
        *reinterpret_cast<const float4 *>(Wp + (tid + q * kNNThreads) * 4);
  if (gfast > 0) {
    // (all four loads before the first LDS write: through load_group's run-time choice they compiled to
    // load -> s_waitcnt vmcnt(0) -> ds_write four times over - four dependent HBM round trips, most of the
    // 8.7 us a block spent in its prologue, tools/probe/nn_stamps.py)
    const float4 t0 = load_fast(0, 0), t1 = load_fast(0, 1), t2 = load_fast(0, 2), t3 = load_fast(0, 3);
    *reinterpret_cast<float4 *>(As + a_lds) = t0;
    *reinterpret_cast<float4 *>(As + a_lds + 32 * ALD) = t1;
    *reinterpret_cast<float4 *>(As + a_lds + 64 * ALD) = t2;
    *reinterpret_cast<float4 *>(As + a_lds + 96 * ALD) = t3;
  } else {
    float4 x1[4], x2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) load_slow(0, i, x1[i], x2[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<float4 *>(As + a_lds + 32 * i * ALD) = pick_slow(0, x1[i], x2[i]);
  }
  __syncthreads();
  RM_STAMP(1);
#ifdef RM_NN_STAMP
  const uint64_t clk0 = __builtin_amdgcn_s_memtime();
#endif

  // B operands of one k-step: the lane's NT weights = two ds_read_b128
  auto read_b = [&](const float *Wb, int s, float4 &t0, float4 &t1) {
    const int k = 8 * (s >> 2) + 4 * h + (s & 3);
    const float *wp = Wb + k * 256 + c * 4;
    t0 = *reinterpret_cast<const float4 *>(wp);
    if constexpr (NT > 4) t1 = *reinterpret_cast<const float4 *>(wp + 128);
  };
  // the next A group, in flight over the two chunks of the current one (named registers: arrays
  // captured by the lambda went to scratch); b*: the slow loader's A2 candidates
  float4 an0, an1, an2, an3, bn0, bn1, bn2, bn3;
  // one weight chunk: prefetch the next chunk (always - the last iteration re-loads its own chunk
  // into the idle buffer: a conditional prefetch made hipcc park the registers in scratch behind
  // an s_waitcnt vmcnt(0)), KC/2 k-steps of MFMAs, commit the prefetch, barrier.
  // first_of_group: also issue the next A group's loads; otherwise commit them before the barrier.
  auto chunk = [&](int ch, auto first_of_group, auto fast_loader) {
    constexpr bool kFirst = decltype(first_of_group)::value;
    constexpr bool kFast = decltype(fast_loader)::value;
    const int nx = ch + 1 < a.nch ? ch + 1 : ch;
    const float *wsrc = Wp + (int64_t)nx * WCHG + tid * 4;
#define RM_W(q) \
  const float4 w##q = !(RM_NN_ABL & 4) ? *reinterpret_cast<const float4 *>(wsrc + q * kNNThreads * 4) : float4{}
    RM_W(0); RM_W(1); RM_W(2); RM_W(3);
#undef RM_W
    const int g = ch >> 1;
    const int gn = g + 1 < ngroups ? g + 1 : g;
    if constexpr (kFirst && !(RM_NN_ABL & 8)) {
      if constexpr (kFast) {
        an0 = load_fast(gn, 0); an1 = load_fast(gn, 1); an2 = load_fast(gn, 2); an3 = load_fast(gn, 3);
      } else {
        load_slow(gn, 0, an0, bn0); load_slow(gn, 1, an1, bn1);
        load_slow(gn, 2, an2, bn2); load_slow(gn, 3, an3, bn3);
      }
    }
    // keep the prefetch HERE: without the fence the scheduler sinks the global loads to their
    // first use (the ds_write at the end of the chunk) and the whole L2 latency is exposed
    __builtin_amdgcn_sched_barrier(0);
    const float *Wb = Ws + (ch & 1) * WCHG;
    // this lane's A fragments of the chunk: row 32*rg + c, k = 16*(ch & 1) + 8q + 4h + e
    const float *ap = As + (g & 1) * (kNNRows * ALD) + (32 * rg + c) * ALD + 16 * (ch & 1) + 4 * h;
    float4 acur[AQ];
#pragma unroll
    for (int q = 0; q < AQ; ++q) acur[q] = *reinterpret_cast<const float4 *>(ap + 8 * q);
    float4 b0[2], b1[2];
    read_b(Wb, 0, b0[0], b1[0]);
#if RM_GEMM_SGB
    // the schedule groups are filled in PROGRAM order: this one takes the chunk's opening reads (A
    // fragments + step 0's weights), so that inside the loop [reads of step s + 1] precede [MFMAs of
    // step s].  (Measured neutral-to-slower on this kernel: profiles/r01_p11.)
    __builtin_amdgcn_sched_group_barrier(0x100, AQ + (NT > 4 ? 2 : 1), 0);
#endif
#pragma unroll
    for (int s = 0; s < KC / 2; ++s) {
      if (s + 1 < KC / 2 && !(RM_NN_ABL & 16)) read_b(Wb, s + 1, b0[(s + 1) & 1], b1[(s + 1) & 1]);
      if (RM_NN_ABL & 16) { b0[(s + 1) & 1] = b0[s & 1]; b1[(s + 1) & 1] = b1[s & 1]; }
      const float4 aq = acur[s >> 2];
      const float av = (s & 3) == 0 ? aq.x : ((s & 3) == 1 ? aq.y : ((s & 3) == 2 ? aq.z : aq.w));
      const float4 t0 = b0[s & 1], t1 = b1[s & 1];
      const float bv[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[nt], acc[nt], 0, 0, 0);
#if RM_GEMM_SGB
      if (s + 1 < KC / 2) __builtin_amdgcn_sched_group_barrier(0x100, NT > 4 ? 2 : 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);
#endif
    }
#if RM_GEMM_SGB
    __builtin_amdgcn_sched_barrier(0);  // the commit of the prefetch stays BEHIND the chunk's MFMAs
#else
    // (slow loader: hipcc hoisted pick_slow's selects - and their s_waitcnt vmcnt - above the MFMAs)
    if constexpr (!kFast && !kFirst) __builtin_amdgcn_sched_barrier(0);
#endif
    float *wdst = Ws + ((ch + 1) & 1) * WCHG + tid * 4;
#define RM_W(q) \
  if constexpr (!(RM_NN_ABL & 32)) *reinterpret_cast<float4 *>(wdst + q * kNNThreads * 4) = w##q
    RM_W(0); RM_W(1); RM_W(2); RM_W(3);
#undef RM_W
    if constexpr (!kFirst && !(RM_NN_ABL & 8)) {  // the group is done after this chunk: publish the next one
      float *adst = As + ((g + 1) & 1) * (kNNRows * ALD) + a_lds;
      if constexpr (kFast) {
        *reinterpret_cast<float4 *>(adst) = an0;
        *reinterpret_cast<float4 *>(adst + 32 * ALD) = an1;
        *reinterpret_cast<float4 *>(adst + 64 * ALD) = an2;
        *reinterpret_cast<float4 *>(adst + 96 * ALD) = an3;
      } else {
        *reinterpret_cast<float4 *>(adst) = pick_slow(gn, an0, bn0);
        *reinterpret_cast<float4 *>(adst + 32 * ALD) = pick_slow(gn, an1, bn1);
        *reinterpret_cast<float4 *>(adst + 64 * ALD) = pick_slow(gn, an2, bn2);
        *reinterpret_cast<float4 *>(adst + 96 * ALD) = pick_slow(gn, an3, bn3);
      }
    }
    if constexpr (!(RM_NN_ABL & 2)) __syncthreads();
  };
  static_assert(KC == 16, "two weight chunks per A group");
  int ch = 0;
  for (; ch + 1 < a.nch && (ch >> 1) + 1 < gfast; ch += 2) {  // groups whose successor is a fast group
    chunk(ch, std::true_type{}, std::true_type{});
    chunk(ch + 1, std::false_type{}, std::true_type{});
  }
  for (; ch + 1 < a.nch; ch += 2) {
    chunk(ch, std::true_type{}, std::false_type{});
    chunk(ch + 1, std::false_type{}, std::false_type{});
  }
  if (ch < a.nch) chunk(ch, std::true_type{}, std::false_type{});  // odd chunk count: the last group has one chunk
  RM_STAMP(2);
#ifdef RM_NN_STAMP
  if (tid == 0 && blockIdx.x < 8192) rm_nn_stamp_buf[8 * blockIdx.x + 7] = __builtin_amdgcn_s_memtime() - clk0;  // shader clocks
#endif
  // ---- epilogue.  The f32 MFMA runs on the SIMD's vector ALU: beside a partner block in its chunk loop a
  // VALU instruction of this wave gets one issue slot per partner MFMA (64 cycles), whatever its priority
  // (s_setprio(3) changed nothing) - the first version's epilogue (per element: 64-bit address arithmetic,
  // bounds predicate with exec-mask updates, bias add, activation) took 30-40 us beside a computing partner
  // and 10-18 us beside one in its own epilogue (tools/probe/nn_stamps.py).  So: as few vector instructions
  // per element as possible.  Every access is a raw buffer access on a tile-based descriptor: rows past M
  // fall outside num_records (loads return 0, stores are dropped), lanes whose column is past N get an
  // out-of-range offset, the row term of the address is a scalar offset (SALU) - no predicates, no 64-bit
  // vector arithmetic; the bias is already in the accumulators.
  // vmcnt retires loads and stores in ONE order: a load issued behind stores is not usable before those
  // stores are acknowledged - aux values are requested two column blocks at a time, ahead of their stores.
  const int lane_row = 4 * h;  // + 32 rg (in the descriptor base) + (r & 3) + 8 (r >> 2) (scalar offset)
  auto tile_rsrc = [&](const float *p, int64_t ld, bool present = true) -> rsrc_t {
    const int64_t rows_w = a.M - row0 < 32 ? a.M - row0 : 32;  // rows of this wave's row group that exist
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p + row0 * ld), 0,
                                             (present && rows_w > 0) ? (int)((rows_w - 1) * ld + a.N) * 4 : 0, 0x00020000);
  };
  auto run = [&](auto epi_tag, auto act_tag) {
    constexpr int EPI = decltype(epi_tag)::value;
    constexpr int ACT = decltype(act_tag)::value;  // compile-time: a run-time id cost 4-6 instructions per element
    constexpr bool kAux = EPI != RM_DENSE_BIAS_ACT;
    constexpr int kAuxB = EPI == RM_DENSE_CROSS ? 1 : 2;  // column blocks per batch of aux loads
    const rsrc_t rc = tile_rsrc(a.C, a.ldc);
    // (an absent C2 gets an empty descriptor: its stores are dropped; ADD without aux1 never gets here -
    // the host turns it into BIAS_ACT / identity)
    const rsrc_t rc2 = tile_rsrc(EPI == RM_DENSE_CROSS && a.C2 != nullptr ? a.C2 : a.C, EPI == RM_DENSE_CROSS ? a.ldc2 : a.ldc,
                                 a.C2 != nullptr);
    const rsrc_t r1 = tile_rsrc(kAux ? a.aux1 : a.C, kAux ? a.ld1 : a.ldc);
    const rsrc_t r2 = tile_rsrc(EPI == RM_DENSE_CROSS ? a.aux2 : a.C, EPI == RM_DENSE_CROSS ? a.ld2 : a.ldc);
    constexpr int kOob = 0x7ffffff0;
    // per-lane offsets (bytes) of column block nt in each array; out of range when the column is past N
    auto lane_off = [&](int nt, int64_t ld) -> int {
      const int col = ct * kTileCols + 32 * (cg == 0 ? nt : tl.nt0 + nt) + c;
      return col < a.N ? (lane_row * (int)ld + col) * 4 : kOob;
    };
    auto row_off = [&](int r, int64_t ld) -> int { return ((r & 3) + 8 * (r >> 2)) * (int)ld * 4; };
#pragma unroll
    for (int n0 = 0; n0 < NT; n0 += kAuxB) {
      float x1[kAux ? kAuxB : 1][16], x2[EPI == RM_DENSE_CROSS ? kAuxB : 1][16];
      if constexpr (kAux) {
#pragma unroll
        for (int q = 0; q < kAuxB; ++q) {
          const int nt = n0 + q;
          if (nt >= NT || nt >= ntw) continue;
          const int o1 = lane_off(nt, a.ld1), o2 = EPI == RM_DENSE_CROSS ? lane_off(nt, a.ld2) : 0;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            x1[q][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r1, o1, row_off(r, a.ld1), 0));
            if constexpr (EPI == RM_DENSE_CROSS)
              x2[q][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r2, o2, row_off(r, a.ld2), 0));
          }
        }
      }
#pragma unroll
      for (int q = 0; q < kAuxB; ++q) {
        const int nt = n0 + q;
        if (nt >= NT || nt >= ntw) continue;
        const int oc = lane_off(nt, a.ldc), oc2 = EPI == RM_DENSE_CROSS ? lane_off(nt, a.ldc2) : 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc[nt][r];
          if constexpr (EPI == RM_DENSE_BIAS_ACT) v = act_apply(v, ACT);
          if constexpr (EPI == RM_DENSE_MUL_ACTGRAD) v *= act_grad_from_out(x1[q][r], ACT);
          if constexpr (EPI == RM_DENSE_ADD) v += x1[q][r];
          if constexpr (EPI == RM_DENSE_CROSS) {
            // u, kept for the backward
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), rc2, oc2, row_off(r, a.ldc2), 0);
            v = x1[q][r] * v + x2[q][r];
          }
          if (!(RM_NN_ABL & 1) || v == 1234.5f)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), rc, oc, row_off(r, a.ldc), 0);
        }
      }
    }
  };
  auto run_act = [&](auto epi_tag) {
    if (a.act == RM_ACT_RELU) run(epi_tag, std::integral_constant<int, RM_ACT_RELU>{});
    else if (a.act == RM_ACT_LEAKY_RELU) run(epi_tag, std::integral_constant<int, RM_ACT_LEAKY_RELU>{});
    else run(epi_tag, std::integral_constant<int, RM_ACT_IDENTITY>{});
  };
  if (a.epi == RM_DENSE_BIAS_ACT) run_act(std::integral_constant<int, RM_DENSE_BIAS_ACT>{});
  else if (a.epi == RM_DENSE_MUL_ACTGRAD) run_act(std::integral_constant<int, RM_DENSE_MUL_ACTGRAD>{});
  else if (a.epi == RM_DENSE_ADD) run(std::integral_constant<int, RM_DENSE_ADD>{}, std::integral_constant<int, RM_ACT_IDENTITY>{});
  else run(std::integral_constant<int, RM_DENSE_CROSS>{}, std::integral_constant<int, RM_ACT_IDENTITY>{});
#ifdef RM_NN_STAMP
  __builtin_amdgcn_s_waitcnt(0);  // (the stores have left the wave)
#endif
  RM_STAMP(3);
}

// One block = 4 waves (one per SIMD) = 128 batch rows x ONE column group of <= 7 32-column blocks; TWO
// blocks per CU.  P0 / P1: accumulators of column group 0 / 1 of a column tile: (7,6) for a 13-block
// N = 400, (7,7) for 14 blocks, (n,n) otherwise.  The two groups of a row tile used to be the two waves
// of each SIMD inside ONE 8-wave block: every barrier, staging phase and epilogue then stopped the MFMA
// pipes of the whole CU at once (MfmaUtil 0.63, profiles/r02_dense_gemm.md).  As independent blocks they
// drift out of phase and one's MFMAs cover the other's barrier / staging / epilogue; the weights are
// still staged once per 128 rows (each block stages only its own group's half), only the A tile is
// read by both (the second read hits L2: blocks w and w + span run on the same XCD).
template <int P0, int P1>
__global__ __launch_bounds__(kNNThreads, 2) void dense_nn_kernel(NNArgs a) {
  extern __shared__ __attribute__((aligned(16))) float Ws[];  // [2][WCHG] + [2][128][36]
  const int tid = threadIdx.x, lane = tid & 63;
  const int rg = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform -> SGPR
  const int c = lane & 31, h = lane >> 5;
  const int per_round = a.span * a.groups;
  // (integer division runs on the VALU: without readfirstlane hipcc treats the quotients as divergent and
  // wraps every buffer load that uses a tile-derived descriptor in a waterfall loop)
  const int w = blockIdx.x, round = __builtin_amdgcn_readfirstlane(w / per_round), in_round = w - round * per_round;
  const int gi = __builtin_amdgcn_readfirstlane(in_round / a.span);
  const int64_t tile = (int64_t)round * a.span + (in_round - gi * a.span);
  if (tile * kNNRows >= a.M) return;
  const int ct = gi >> 1, cg = gi & 1;
  // Phase stagger.  The two blocks of a CU start together and do the same work at the same speed: left
  // alone they stay in lockstep (tools/probe/nn_stamps.py: both in the chunk loop for 87 us, then both in
  // the epilogue for 10-20 us - and so is every other CU: the whole chip computes with HBM idle, then
  // stores 54 MB at once with the MFMA pipes idle).  The odd-group blocks of the FIRST round start a little
  // later (5 % of a block, ~4 us: enough to take the two epilogues apart; 0 / 5 / 10 / 25 % all run the same
  // speed, profiles/r02_dense_gemm.md); every later round inherits the offset because a CU slot is refilled
  // the moment its block ends.
  if (a.stagger_ticks > 0 && round == 0 && cg == 1) {
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (uint64_t)a.stagger_ticks) __builtin_amdgcn_s_sleep(16);
  }
  const ColTile tl = col_tile(a.N, ct);
  if (cg == 0) nn_wave<P0>(a, Ws, tid, c, h, rg, 0, ct, tl, tl.nt0, tile);
  else if (tl.nb > tl.nt0) nn_wave<P1>(a, Ws, tid, c, h, rg, 1, ct, tl, tl.nb - tl.nt0, tile);
}

// ---------------------------------------------------------------------------------------------
struct TNArgs {
  const float *A1, *A2;
  int64_t lda1, lda2;
  int K1, K2;
  const float *G;
  int64_t ldg;
  int N;
  int64_t M;
  int slabs, kts;      // batch slabs, 128-row tiles of K
  int kts_fast;        // K tiles [0, kts_fast) lie inside A1 with 16-byte rows (float4 staging)
  int a_vec;           // A1 rows are 16-byte aligned and K1 % 4 == 0
  // Ragged last K tile with 1 or 2 live row groups (K = 400: 16 rows, K = 429: 45): its blocks would
  // keep one or two SIMDs busy and idle the rest for a whole slab.  With split_f = 4 / live > 1 the
  // tile's 4 row-group slots are (live row group) x (split_f parts of every chunk's 16 k-steps), the
  // tile gets slabs_r (< slabs) longer slabs, and each part stores its own partial slab.
  int split_f, live_rg, slabs_r;
  int64_t slab_rows_r;
  int64_t slab_rows;   // multiple of TRC
  float *ws;           // [slabs][K][N]
  float *db_ws;        // [slabs][N] column sums of G per slab (the bias gradient), or NULL
};

// TN: one block = 128 rows of K (4 row groups) x one column tile, one slab of the batch.  Per
// chunk of 32 batch rows the A piece [32][128] and the G piece [32][cols] are staged in LDS with
// wide coalesced loads (ONE global read per block; the first version let every wave load its
// operands straight from global - 4x / 2x redundant dword loads, 70 % of the wave time in
// s_waitcnt vmcnt) and all 8 waves read their MFMA operands from there (conflict-free b32).
constexpr int TRC = 32;                         // batch rows per chunk
constexpr int TLDA = kRowsPerBlock + 4;         // LDS row stride of the A piece (floats)
constexpr int TLDG = kTileCols + 4;             // LDS row stride of the G piece
constexpr int kTnSmemFloats = TRC * (TLDA + TLDG);
constexpr int TAQ = TRC * kRowsPerBlock / 4 / kThreads;          // float4 of A per thread per chunk (2)
constexpr int TGQ = (TRC * kTileCols / 4 + kThreads - 1) / kThreads;  // float4 of G per thread (7)

// k-steps of one staged chunk; the B operands of step s+1 are read before the MFMAs of step s
template <int NT, int STEPS = TRC / 2>
__device__ __forceinline__ void tn_compute(const float *As, const float *Gs, f32x16 (&acc)[NT], int c,
                                           int h, int rg, int colb, int s0 = 0) {
  const float *ap = As + (2 * s0 + h) * TLDA + 32 * rg + c;
  const float *gp = Gs + (2 * s0 + h) * TLDG + colb + c;
  float av[2], bv[2][NT];
  av[0] = ap[0];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bv[0][nt] = gp[32 * nt];
#pragma unroll
  for (int s = 0; s < STEPS; ++s) {
    if (s + 1 < STEPS) {
      av[(s + 1) & 1] = ap[(2 * s + 2) * TLDA];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) bv[(s + 1) & 1][nt] = gp[(2 * s + 2) * TLDG + 32 * nt];
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s & 1], bv[s & 1][nt], acc[nt], 0, 0, 0);
    // "next step's LDS reads, then this step's MFMAs" (left alone, hipcc read two operands,
    // waited, issued two MFMAs, ... - an exposed LDS latency per pair of MFMAs)
    __builtin_amdgcn_sched_group_barrier(0x100, NT + 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);
  }
}

// The whole wave program with exactly NT accumulators (both column groups run the same number
// of barriers); colb = first column of this wave's group inside the tile.
// FASTA: the block's A piece lies inside A1 with 16-byte rows; FASTG: G rows are 16-byte aligned
// and N % 4 == 0 (float4 staging, 32-bit offsets from scalar row pointers); otherwise
// per-element loads (the ragged last K tile: x = [A1 | A2] boundary and the end of K).
// Staged values outside [K] x [N] need no zeroing: an MFMA output row / column depends only on
// its own A row / B column, and those outputs are never stored.  Batch rows past the end ARE
// zeroed (peeled last chunk).
template <int NT, bool FASTA, bool FASTG, int F = 1>
__device__ __forceinline__ void tn_wave(const TNArgs &a, float *smem, int tid, int c, int h, int rg_slot,
                                        int ntw, int colb, int slab, int kt) {
  float *As = smem;                 // [TRC][TLDA]
  float *Gs = smem + TRC * TLDA;    // [TRC][TLDG]
  const int ct = blockIdx.y;
  const int K = a.K1 + a.K2;
  const int ka0 = kt * kRowsPerBlock, col0 = ct * kTileCols;
  const int ncols = min(a.N - col0, kTileCols);  // real columns of this tile
  // F > 1: the split ragged tile - slot -> (live row group, part of the chunk's k-steps)
  const int rg = F > 1 ? rg_slot % a.live_rg : rg_slot;
  const int part = F > 1 ? rg_slot / a.live_rg : 0;
  const int64_t srows = F > 1 ? a.slab_rows_r : a.slab_rows;
  const int64_t b_begin = (int64_t)slab * srows;
  int64_t b_end = b_begin + srows;
  b_end = b_end < a.M ? b_end : a.M;

  // per-thread staging coordinates (constant over the chunks)
  uint32_t offA[TAQ], offG[TGQ];
  int ldsA[TAQ], ldsG[TGQ];
#pragma unroll
  for (int q = 0; q < TAQ; ++q) {
    const int f = tid + q * kThreads, bl = f / (kRowsPerBlock / 4), k4 = f % (kRowsPerBlock / 4);
    offA[q] = (uint32_t)(bl * (FASTA ? a.lda1 : 0)) + (uint32_t)(ka0 + 4 * k4);
    ldsA[q] = bl * TLDA + 4 * k4;
  }
#pragma unroll
  for (int q = 0; q < TGQ; ++q) {
    const int f = tid + q * kThreads, bl = f / (kTileCols / 4), c4 = f % (kTileCols / 4);
    const int n = 4 * c4 < ncols ? 4 * c4 : 0;  // columns past the tile: any in-bounds address
    offG[q] = (uint32_t)(bl * a.ldg) + (uint32_t)(col0 + n);
    ldsG[q] = bl * TLDG + 4 * c4;
  }
  const float *A2 = a.K2 > 0 ? a.A2 : a.A1;
  const int64_t lda2 = a.K2 > 0 ? a.lda2 : a.lda1;

  // staged values live in NAMED registers (float4 arrays captured by a lambda went to scratch)
  static_assert(TAQ == 2 && TGQ == 7, "the staging below is written out for 2 + 7 float4 per thread");
  float4 pa0, pa1, pg0, pg1, pg2, pg3, pg4, pg5, pg6;
  auto load_a = [&](int q, uint32_t off, int64_t r0, int rows_left, bool full) -> float4 {
    const int bl = (tid + q * kThreads) / (kRowsPerBlock / 4);
    const bool rok = full || bl < rows_left;
    float4 v;
    if constexpr (FASTA) {
      const float *Ap = a.A1 + r0 * a.lda1;  // wave-uniform row pointer
      v = *reinterpret_cast<const float4 *>(Ap + (rok ? off : off - bl * (uint32_t)a.lda1));
    } else {
      // ragged K tile: columns from A1, from the second piece A2, or past the end of K (any
      // finite value will do there - those output rows are never stored)
      const int64_t b = r0 + (rok ? bl : 0);
      const int k = (int)off;
      float t1[4], t2[4];
      if (a.a_vec) {  // A1 rows 16-byte aligned and K1 % 4 == 0: one float4 + 4 dwords of A2
        const int kc = k < a.K1 ? k : a.K1 - 4;
        const float4 v1 = *reinterpret_cast<const float4 *>(a.A1 + b * a.lda1 + kc);
        t1[0] = v1.x; t1[1] = v1.y; t1[2] = v1.z; t1[3] = v1.w;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) t1[e] = a.A1[b * a.lda1 + (k + e < a.K1 ? k + e : 0)];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k2 = k + e - a.K1;
        t2[e] = A2[b * lda2 + ((k2 >= 0 && k2 < a.K2) ? k2 : 0)];
      }
      float t[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) t[e] = k + e < a.K1 ? t1[e] : t2[e];
      v = make_float4(t[0], t[1], t[2], t[3]);
    }
    return rok ? v : make_float4(0.f, 0.f, 0.f, 0.f);  // batch rows past the end contribute 0
  };
  auto load_g = [&](int q, uint32_t off0, int64_t r0, int rows_left, bool full) -> float4 {
    const int bl = (tid + q * kThreads) / (kTileCols / 4);
    const bool rok = full || bl < rows_left;
    const uint32_t off = rok ? off0 : off0 - bl * (uint32_t)a.ldg;
    const float *Gp = a.G + r0 * a.ldg;
    // (rows past the end of the batch: zero - A's zero rows already keep them out of dW, the
    // column sums for db need G's own)
    if constexpr (FASTG) {
      const float4 v = *reinterpret_cast<const float4 *>(Gp + off);
      return rok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      // per-element path (N % 4 != 0 or unaligned rows - typically a skinny G such as the 7
      // coefficient columns of the cross-net gradients): lanes whose columns lie past the tile
      // load nothing (their LDS slots feed only outputs that are never stored)
      const int c4 = (tid + q * kThreads) % (kTileCols / 4);
      float t[4] = {0.f, 0.f, 0.f, 0.f};
      if (4 * c4 < ncols) {
#pragma unroll
        for (int e = 0; e < 4; ++e) t[e] = Gp[off + ((4 * c4 + e < ncols) ? e : 0)];
      }
      return rok ? make_float4(t[0], t[1], t[2], t[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
#define RM_TN_PREFETCH(r0_, full_)                                                          \
  {                                                                                         \
    const int64_t r0v = (r0_);                                                              \
    const int rl = (int)(b_end - r0v);                                                      \
    pa0 = load_a(0, offA[0], r0v, rl, full_); pa1 = load_a(1, offA[1], r0v, rl, full_);      \
    pg0 = load_g(0, offG[0], r0v, rl, full_); pg1 = load_g(1, offG[1], r0v, rl, full_);      \
    pg2 = load_g(2, offG[2], r0v, rl, full_); pg3 = load_g(3, offG[3], r0v, rl, full_);      \
    pg4 = load_g(4, offG[4], r0v, rl, full_); pg5 = load_g(5, offG[5], r0v, rl, full_);      \
    pg6 = load_g(6, offG[6], r0v, rl, full_);                                               \
  }
#define RM_TN_COMMIT()                                                                      \
  {                                                                                         \
    *reinterpret_cast<float4 *>(As + ldsA[0]) = pa0; *reinterpret_cast<float4 *>(As + ldsA[1]) = pa1; \
    *reinterpret_cast<float4 *>(Gs + ldsG[0]) = pg0; *reinterpret_cast<float4 *>(Gs + ldsG[1]) = pg1; \
    *reinterpret_cast<float4 *>(Gs + ldsG[2]) = pg2; *reinterpret_cast<float4 *>(Gs + ldsG[3]) = pg3; \
    *reinterpret_cast<float4 *>(Gs + ldsG[4]) = pg4; *reinterpret_cast<float4 *>(Gs + ldsG[5]) = pg5; \
    *reinterpret_cast<float4 *>(Gs + ldsG[6]) = pg6;                                        \
  }

  f32x16 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
  const bool rg_live = ka0 + 32 * rg < K;  // a row group past the end of K only helps staging

  // db (optional): the column sums of G ride along in the K-tile-0 blocks - the G piece of every
  // chunk is in LDS anyway, thread t adds up column t of it (a pass over G less than a separate
  // bias-gradient kernel: 45 us at DCN's shape).  Block-uniform.
  const bool do_db = a.db_ws != nullptr && kt == 0;
  const int dbc = tid < kTileCols ? tid : 0;
  float csum = 0.f;
  auto colsum = [&]() {
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int bl = 0; bl < TRC; bl += 2) {
      s0 += Gs[bl * TLDG + dbc];
      s1 += Gs[(bl + 1) * TLDG + dbc];
    }
    csum += s0 + s1;
  };

  const int64_t nrows = b_end > b_begin ? b_end - b_begin : 0;
  const int64_t nfull = nrows / TRC;  // full chunks; a partial one may follow
  const bool partial = nrows % TRC != 0;
  if (nfull > 0) RM_TN_PREFETCH(b_begin, true)
  for (int64_t ci = 0; ci < nfull; ++ci) {
    __syncthreads();  // previous chunk fully consumed
    RM_TN_COMMIT()
    __syncthreads();
    // the next chunk's global loads fly during this chunk's MFMAs (unconditional: the last
    // iteration re-loads its own chunk - a conditional prefetch parks the registers in scratch)
    const int64_t nx = ci + 1 < nfull ? ci + 1 : ci;
    RM_TN_PREFETCH(b_begin + nx * TRC, true)
    __builtin_amdgcn_sched_barrier(0);
    if (do_db) colsum();
    if (rg_live) tn_compute<NT, TRC / 2 / F>(As, Gs, acc, c, h, rg, colb, part * (TRC / 2 / F));
  }
  if (partial) {  // the batch's ragged end (last slab only): rows past the end staged as zeros
    RM_TN_PREFETCH(b_begin + nfull * TRC, false)
    __syncthreads();
    RM_TN_COMMIT()
    __syncthreads();
    if (do_db) colsum();
    if (rg_live) tn_compute<NT, TRC / 2 / F>(As, Gs, acc, c, h, rg, colb, part * (TRC / 2 / F));
  }
  if (do_db && tid < ncols) a.db_ws[(int64_t)slab * a.N + col0 + tid] = csum;
  if (!rg_live) return;
  float *out = a.ws + (int64_t)(F > 1 ? slab * F + part : slab) * K * a.N;  // (a part = a partial slab)
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int col = col0 + colb + 32 * nt + c;
    if (nt >= ntw || col >= a.N) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kr = ka0 + 32 * rg + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (kr < K) out[(int64_t)kr * a.N + col] = acc[nt][r];
    }
  }
}

#undef RM_TN_PREFETCH
#undef RM_TN_COMMIT

template <int P0, int P1, bool FASTG>
__global__ __launch_bounds__(kThreads, 1) void dense_tn_kernel(TNArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 31, h = lane >> 5, rg = wave & 3, cg = wave >> 2;
  const ColTile tl = col_tile(a.N, blockIdx.y);
  // XCD-aware order: the K tiles of one slab run next to each other on ONE XCD (they re-read
  // the same G rows: 3 of 4 reads then hit that XCD's L2)
  int64_t L = blockIdx.x;
  // the K tiles with `slabs` slabs each: all of them, or all but a split ragged last one
  const int ktsf = a.split_f > 1 ? a.kts - 1 : a.kts;
  const int64_t total = (int64_t)a.slabs * ktsf;
  const int ntw = cg == 0 ? tl.nt0 : tl.nb - tl.nt0, colb = cg == 0 ? 0 : 32 * tl.nt0;
  if (L >= total) {  // the split ragged tile's blocks (block-uniform)
    const int slab = (int)(L - total), kt = a.kts - 1;
    if (a.split_f == 4) {
      if (cg == 0) tn_wave<P0, false, FASTG, 4>(a, smem, tid, c, h, rg, ntw, colb, slab, kt);
      else tn_wave<P1, false, FASTG, 4>(a, smem, tid, c, h, rg, ntw, colb, slab, kt);
    } else {
      if (cg == 0) tn_wave<P0, false, FASTG, 2>(a, smem, tid, c, h, rg, ntw, colb, slab, kt);
      else tn_wave<P1, false, FASTG, 2>(a, smem, tid, c, h, rg, ntw, colb, slab, kt);
    }
    return;
  }
  {  // blocks L, L + 8, L + 16, ... share an XCD: give XCD x a contiguous range of (slab, K tile) pairs
    const int x = (int)(L & 7);
    int64_t first = 0;
    for (int y = 0; y < x; ++y) first += (total - y + 7) >> 3;
    L = first + (L >> 3);
  }
  const int slab = (int)(L / ktsf), kt = (int)(L % ktsf);
  const bool fa = kt < a.kts_fast;  // block-uniform: this K tile lies inside A1 with 16-byte rows
  if (cg == 0) {
    if (fa) tn_wave<P0, true, FASTG>(a, smem, tid, c, h, rg, ntw, colb, slab, kt);
    else tn_wave<P0, false, FASTG>(a, smem, tid, c, h, rg, ntw, colb, slab, kt);
  } else {
    if (fa) tn_wave<P1, true, FASTG>(a, smem, tid, c, h, rg, ntw, colb, slab, kt);
    else tn_wave<P1, false, FASTG>(a, smem, tid, c, h, rg, ntw, colb, slab, kt);
  }
}

// dW[k][n] (+)= sum_slab ws[slab][k][n].  A block owns 64 consecutive outputs, its 8 waves take every
// 8th slab (a thread's chain of loads is slabs / 8 long), fixed-order tree in LDS -> deterministic.
constexpr int kTrG = 8;
__global__ __launch_bounds__(64 * kTrG) void dense_tn_reduce(const float *__restrict__ ws, int slabs, int64_t KN,
                                                            int N, float *dW, int64_t lddw, int accumulate,
                                                            const float *__restrict__ db_ws,
                                                            float *__restrict__ db, int64_t t_hi,
                                                            int slabs_hi, int slabs_db) {
  // elements t >= t_hi (the rows of a split ragged K tile) have slabs_hi partial slabs instead
  if (t_hi <= 0) { t_hi = KN; slabs_hi = slabs; }
  if (slabs_db <= 0) slabs_db = slabs;
  __shared__ float sm[kTrG][64];
  const int o = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int64_t total = KN + (db != nullptr ? N : 0);  // the db columns ride behind the dW elements
  for (int64_t t0 = (int64_t)blockIdx.x * 64; t0 < total; t0 += (int64_t)gridDim.x * 64) {
    const int64_t t = t0 + o;
    float s = 0.f;
    if (t < KN) {
      const int sl = t < t_hi ? slabs : slabs_hi;
#pragma unroll 4
      for (int q = grp; q < sl; q += kTrG) s += ws[(int64_t)q * KN + t];
    } else if (t < total) {
#pragma unroll 4
      for (int q = grp; q < slabs_db; q += kTrG) s += db_ws[(int64_t)q * N + (t - KN)];
    }
    sm[grp][o] = s;
    __syncthreads();
    if (grp == 0 && t < total) {
      float v[kTrG];
#pragma unroll
      for (int q = 0; q < kTrG; ++q) v[q] = sm[q][o];
#pragma unroll
      for (int st = 1; st < kTrG; st *= 2)
#pragma unroll
        for (int q = 0; q < kTrG; q += 2 * st) v[q] += v[q + st];
      if (t < KN) {
        const int64_t k = t / N, n = t - k * N;
        float *d = dW + k * lddw + n;
        *d = accumulate ? *d + v[0] : v[0];
      } else {
        db[t - KN] = v[0];
      }
    }
    __syncthreads();
  }
}

// Skinny G (N <= 8: the cross-net's coefficient columns): dW[k][n] = sum_b x[b][k] G[b][n] is a
// handful of weighted column sums of x - HBM-bound (x read once), not MFMA work: the tiled TN kernel
// above padded N to 32 columns and spent its time in staging barriers (166 us at DCN's shape, 1.4 TB/s).
// A block owns rows_per_block rows; G's rows of the block sit in LDS (every thread needs all N of a
// row: broadcast reads); a thread owns one float4 column group of A1 or one column of A2 and keeps
// its N x 4 partial sums in registers.  One partial [K][N] per block -> dense_tn_reduce.
constexpr int kSkN = 8;       // max N of this path
constexpr int kSkRows = 128;  // rows per block
__global__ __launch_bounds__(128) void dense_tn_skinny_kernel(
    const float *__restrict__ A1, int64_t lda1, int K1, const float *__restrict__ A2, int64_t lda2, int K2,
    const float *__restrict__ G, int64_t ldg, int N, int64_t M, float *__restrict__ ws) {
  __shared__ float Gs[kSkRows][kSkN];
  const int tid = threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.x * kSkRows;
  const int nrows = (int)((M - r0) < kSkRows ? (M - r0) : kSkRows);
  for (int t = tid; t < kSkRows * kSkN; t += 128) {
    const int r = t / kSkN, n = t - r * kSkN;
    Gs[r][n] = (r < nrows && n < N) ? G[(r0 + r) * ldg + n] : 0.f;
  }
  __syncthreads();
  const int K = K1 + K2, K14 = K1 / 4;       // (host: K1 % 4 == 0 and 16-byte rows)
  float *out = ws + (int64_t)blockIdx.x * K * N;
  for (int item = tid; item < K14 + K2; item += 128) {
    float acc[kSkN][4];
#pragma unroll
    for (int n = 0; n < kSkN; ++n)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[n][e] = 0.f;
    if (item < K14) {
      const float *ap = A1 + r0 * lda1 + 4 * item;
#pragma unroll 4
      for (int r = 0; r < nrows; ++r) {
        const float4 x = *reinterpret_cast<const float4 *>(ap + (int64_t)r * lda1);
        const float4 g0 = *reinterpret_cast<const float4 *>(&Gs[r][0]);
        const float4 g1 = *reinterpret_cast<const float4 *>(&Gs[r][4]);
        const float gv[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
        for (int n = 0; n < kSkN; ++n) {
          acc[n][0] += gv[n] * x.x; acc[n][1] += gv[n] * x.y;
          acc[n][2] += gv[n] * x.z; acc[n][3] += gv[n] * x.w;
        }
      }
      for (int n = 0; n < N; ++n)
#pragma unroll
        for (int e = 0; e < 4; ++e) out[(int64_t)(4 * item + e) * N + n] = acc[n][e];
    } else {
      const int k2 = item - K14;
      const float *ap = A2 + r0 * lda2 + k2;
#pragma unroll 4
      for (int r = 0; r < nrows; ++r) {
        const float x = ap[(int64_t)r * lda2];
        const float4 g0 = *reinterpret_cast<const float4 *>(&Gs[r][0]);
        const float4 g1 = *reinterpret_cast<const float4 *>(&Gs[r][4]);
        const float gv[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
        for (int n = 0; n < kSkN; ++n) acc[n][0] += gv[n] * x;
      }
      for (int n = 0; n < N; ++n) out[(int64_t)(K1 + k2) * N + n] = acc[n][0];
    }
  }
}

// first reduction stage of the skinny path's many partials: out[y][t] = sum of `group` consecutive
// partials (coalesced over t, a short fixed-order chain per thread); dense_tn_reduce finishes
__global__ void dense_tn_fold_kernel(const float *__restrict__ in, int n_in, int group, int64_t KN,
                                     float *__restrict__ out) {
  const int y = blockIdx.y;
  const int q0 = y * group, q1 = q0 + group < n_in ? q0 + group : n_in;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < KN;
       t += (int64_t)gridDim.x * blockDim.x) {
    float s = 0.f;
#pragma unroll 8
    for (int q = q0; q < q1; ++q) s += in[(int64_t)q * KN + t];
    out[(int64_t)y * KN + t] = s;
  }
}

constexpr int kSkFold = 32;  // partials per first-stage group
bool tn_skinny_ok(int K1, int N) { return N <= kSkN && K1 % 4 == 0; }
int64_t tn_skinny_blocks(int64_t M) { return (M + kSkRows - 1) / kSkRows; }
int64_t tn_skinny_groups(int64_t M) { return (tn_skinny_blocks(M) + kSkFold - 1) / kSkFold; }

int tn_slabs(int64_t M, int kts, int ncts) {
  int64_t s = 256 / ((int64_t)kts * ncts);  // ONE round of blocks over the 256 CUs (1 block per CU)
  const int64_t cap = (M + 511) / 512;            // at least 512 batch rows per slab
  s = s < cap ? s : cap;
  s = s < 1 ? 1 : s;
  if (s >= 8) s = s / 8 * 8;  // multiple of 8 so the XCD remap applies
  return (int)s;
}

// The launch plan of the tiled TN kernel (one place for the workspace query and the launch).
struct TnPlan {
  int kts, nct, slabs, split_f, live, slabs_r, smax;
  int64_t rows_f, rows_r;
};
TnPlan tn_plan(int K, int N, int64_t M, bool allow_split) {
  TnPlan p;
  p.kts = (K + kRowsPerBlock - 1) / kRowsPerBlock;
  p.nct = ((N + 31) / 32 + kMaxNB - 1) / kMaxNB;
  p.slabs = tn_slabs(M, p.kts, p.nct);
  p.split_f = 1;
  p.live = 4;
  p.slabs_r = 0;
  const int rem = K - (p.kts - 1) * kRowsPerBlock;
  const int live = (rem + 31) / 32;
#ifndef RM_TN_SPLIT
#define RM_TN_SPLIT 1
#endif
#ifndef RM_TN_RATIO2
#define RM_TN_RATIO2 0.667
#define RM_TN_RATIO1 0.41
#endif
  // (only with float4 staging of G: with the per-element G loader the blocks are staging-bound and the
  // ragged tile's longer slabs became the critical path - dcn_matrix's N = 429 went 17.3 -> 18.6 ms)
  if (RM_TN_SPLIT && allow_split && p.kts >= 2 && live <= 2) {
    // A ragged-tile block does a half (live = 2) / a quarter (live = 1) of the MFMA work per chunk, but its
    // staging costs more than a full tile's (per-element loader): measured per chunk 10.3k cycles (live 2) /
    // 7.0k (live 1) against 15.5k for a full tile (profiles/r02_dense_gemm.md).  Its slabs are longer by that
    // ratio so that both kinds of block end together: slabs_r = ratio * slabs, with (kts - 1) slabs +
    // slabs_r <= one round of blocks.  (Round 1 used slabs / 2 for both cases: at K = 429 the ragged blocks
    // then ran 1.33x longer than the rest - 510 us against 385 us at K = 400.)
    const double ratio = live == 2 ? RM_TN_RATIO2 : RM_TN_RATIO1;
    const int64_t round_blocks = 256 / p.nct;
    int64_t sf = (int64_t)(round_blocks / (p.kts - 1 + ratio));
    const int64_t cap = (M + 511) / 512;
    sf = sf < cap ? sf : cap;
    int64_t sr = round_blocks - (p.kts - 1) * sf;              // what is left of the round
    const int64_t sr_want = (int64_t)(ratio * sf + 0.999);
    sr = sr < sr_want ? sr : sr_want;
    if (const char *e = getenv("RECMAN_TN_SLABS")) {  // experiments: "slabs,slabs_r"
      int x = 0, y = 0;
      if (sscanf(e, "%d,%d", &x, &y) == 2 && x >= 4 && y >= 1 && (p.kts - 1) * x + y <= round_blocks) { sf = x; sr = y; }
    }
    if (sf >= 4 && sr >= 1) {
      p.slabs = (int)sf;
      p.split_f = 4 / live;
      p.live = live;
      p.slabs_r = (int)sr;
    }
  }
  p.rows_f = ((M + p.slabs - 1) / p.slabs + TRC - 1) / TRC * TRC;
  p.rows_r = p.slabs_r ? ((M + p.slabs_r - 1) / p.slabs_r + TRC - 1) / TRC * TRC : 0;
  p.smax = p.slabs_r * p.split_f > p.slabs ? p.slabs_r * p.split_f : p.slabs;
  return p;
}

}  // namespace

#ifdef RM_NN_STAMP
extern "C" int rm_debug_nn_stamps(unsigned long long *host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(rm_nn_stamp_buf), sizeof(unsigned long long) * n);
}
#endif

extern "C" int64_t rm_dense_filter_workspace(int K, int N) {
  if (K <= 0 || N <= 0) return 0;
  const int nch = (K + KC - 1) / KC;
  const int nct = ((N + 31) / 32 + kMaxNB - 1) / kMaxNB;
  return (int64_t)nct * nch * WCH;
}

extern "C" int rm_dense_fwd(const float *A1, int64_t lda1, int K1, const float *A2, int64_t lda2, int K2,
                            const float *W, int64_t ldw, int w_transposed, int N, const float *bias,
                            int epilogue, int act, const float *aux1, int64_t ld_aux1,
                            const float *aux2, int64_t ld_aux2, int64_t M, float *C, int64_t ldc,
                            float *C2, int64_t ldc2, float *filter_ws, rm_stream_t stream) {
  RM_REQUIRE(M >= 0 && K1 > 0 && K2 >= 0 && N > 0, "rm_dense_fwd: bad sizes");
  if (M == 0) return RM_OK;
  RM_REQUIRE(A1 && W && C && filter_ws, "rm_dense_fwd: NULL argument");
  RM_REQUIRE(K2 == 0 || A2, "rm_dense_fwd: K2 > 0 needs A2");
  RM_REQUIRE(lda1 >= K1, "rm_dense_fwd: lda1 < K1");
  RM_REQUIRE(K2 == 0 || lda2 >= K2, "rm_dense_fwd: lda2 < K2");
  RM_REQUIRE(lda1 < (1 << 22) && lda2 < (1 << 22) && ldc < (1 << 22) && ldc2 < (1 << 22) && ld_aux1 < (1 << 22) &&
                 ld_aux2 < (1 << 22),
             "rm_dense_fwd: row stride too long for the 32-bit tile offsets");
  RM_REQUIRE(rm_aligned16(filter_ws), "rm_dense_fwd: filter_ws must be 16-byte aligned");
  RM_REQUIRE(epilogue >= RM_DENSE_BIAS_ACT && epilogue <= RM_DENSE_CROSS, "rm_dense_fwd: bad epilogue id");
  RM_REQUIRE(act >= RM_ACT_IDENTITY && act <= RM_ACT_LEAKY_RELU, "rm_dense_fwd: bad activation id");
  RM_REQUIRE(epilogue != RM_DENSE_MUL_ACTGRAD || aux1, "rm_dense_fwd: MUL_ACTGRAD needs aux1");
  RM_REQUIRE(epilogue != RM_DENSE_CROSS || (aux1 && aux2), "rm_dense_fwd: CROSS needs aux1 (x0) and aux2 (x_l)");
  RM_REQUIRE(ldc >= N && (!aux1 || ld_aux1 >= N) && (!aux2 || ld_aux2 >= N) && (!C2 || ldc2 >= N),
             "rm_dense_fwd: a leading dimension is smaller than N");
  const int K = K1 + K2;
  // K1 itself may be ragged only when there is no second piece (tail handled by the loader)
  const int nch = (K + KC - 1) / KC;
  const int nbt = (N + 31) / 32;
  const int nct = (nbt + kMaxNB - 1) / kMaxNB;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(dense_prep_kernel, dim3(rm_grid_cap(((int64_t)nct * nch * WCH + 255) / 256, 1024)),
                     dim3(256), 0, st, W, ldw, w_transposed, K, N, nch, nct, filter_ws);
  RM_CHECK_LAUNCH("rm_dense_fwd(prep)");
  // blockIdx.x -> (row tile, column group): rounds of `span` row tiles, inside a round all tiles of group 0,
  // then all of group 1, ...: the groups of one row tile are `span` blocks apart - with span % 8 == 0 on the
  // same XCD (round-robin placement) and, the dispatcher filling CU slots in order, about the same time
  const int64_t ntiles = (M + kNNRows - 1) / kNNRows;
  const int span = (int)(ntiles < 256 ? ntiles : 256);
  const int groups = 2 * nct;
  RM_REQUIRE((ntiles + span - 1) / span * span * groups < ((int64_t)1 << 31), "rm_dense_fwd: grid too large");
  NNArgs a{A1, A2, lda1, lda2, K1, K2, (rm_aligned16(A1) && lda1 % 4 == 0) ? 1 : 0, filter_ws, N, nch, bias, epilogue, act, aux1, aux2,
           ld_aux1, ld_aux2, M, C, C2, ldc, ldc2, span, groups, 0};
  if (epilogue == RM_DENSE_ADD && !aux1) { a.epi = RM_DENSE_BIAS_ACT; a.act = RM_ACT_IDENTITY; }
  if (ntiles * groups > 512) {  // more blocks than CU slots: rounds exist, stagger them
    static const int pct = [] { const char *e = getenv("RECMAN_NN_STAGGER_PCT"); return e ? atoi(e) : 5; }();
    const int nb_first = nbt < kMaxNB ? nbt : kMaxNB;
    // one block-time ~ nch chunks x 8 k-steps x nb MFMAs x 64 cycles at ~2.3 GHz, in 10 ns ticks
    a.stagger_ticks = (int)((int64_t)pct * nch * 8 * nb_first * 64 / 23 / 100);
  }
  const dim3 grid((unsigned)((ntiles + span - 1) / span * span * groups));
  const size_t smem = (2 * WCHG + 2 * kNNRows * 36) * sizeof(float);  // 68 KB: two blocks per CU
  const int nb0 = nbt < kMaxNB ? nbt : kMaxNB;  // blocks of the widest (first) column tile
#define RM_NN(P0, P1)                                                                         \
  {                                                                                         \
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(dense_nn_kernel<P0, P1>),      \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);       \
    hipLaunchKernelGGL((dense_nn_kernel<P0, P1>), grid, dim3(kNNThreads), smem, st, a);     \
  }
  if (nb0 <= 2) RM_NN(1, 1)
  else if (nb0 <= 4) RM_NN(2, 2)
  else if (nb0 <= 6) RM_NN(3, 3)
  else if (nb0 <= 8) RM_NN(4, 4)
  else if (nb0 <= 10) RM_NN(5, 5)
  else if (nb0 <= 12) RM_NN(6, 6)
  else if (nb0 == 13) RM_NN(7, 6)
  else RM_NN(7, 7)
#undef RM_NN
  RM_CHECK_LAUNCH("rm_dense_fwd");
  return RM_OK;
}

extern "C" int64_t rm_dense_wgrad_workspace(int K, int N, int64_t M) {
  if (K <= 0 || N <= 0 || M <= 0) return 0;
  int64_t need = 0;
  for (int sp = 0; sp < 2; ++sp) {  // whichever plan the call takes (the split needs aligned G rows)
    const TnPlan pl = tn_plan(K, N, M, sp != 0);
    const int64_t n = (int64_t)pl.smax * K * N + (int64_t)pl.slabs * N;  // + the per-slab column sums (db)
    need = n > need ? n : need;
  }
  if (N <= kSkN) {  // the skinny path's one partial per 128 rows (whichever path the call takes)
    const int64_t sk = (tn_skinny_blocks(M) + tn_skinny_groups(M)) * (int64_t)K * N;
    need = sk > need ? sk : need;
  }
  return need;
}

extern "C" int rm_dense_wgrad(const float *A1, int64_t lda1, int K1, const float *A2, int64_t lda2,
                              int K2, const float *G, int64_t ldg, int N, int64_t M, float *dW,
                              int64_t lddw, int accumulate, float *db, float *workspace,
                              int64_t workspace_floats, rm_stream_t stream) {
  RM_REQUIRE(M >= 0 && K1 > 0 && K2 >= 0 && N > 0, "rm_dense_wgrad: bad sizes");
  RM_REQUIRE(A1 && G && dW && workspace, "rm_dense_wgrad: NULL argument");
  RM_REQUIRE(K2 == 0 || A2, "rm_dense_wgrad: K2 > 0 needs A2");
  RM_REQUIRE(lda1 >= K1 && (K2 == 0 || lda2 >= K2) && ldg >= N && lddw >= N,
             "rm_dense_wgrad: a leading dimension is too small");
  const int K = K1 + K2;
  RM_REQUIRE(workspace_floats >= rm_dense_wgrad_workspace(K, N, M > 0 ? M : 1),
             "rm_dense_wgrad: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  if (M > 0 && !db && tn_skinny_ok(K1, N) && lda1 % 4 == 0 && rm_aligned16(A1)) {
    const int nblk = (int)tn_skinny_blocks(M);
    hipLaunchKernelGGL(dense_tn_skinny_kernel, dim3(nblk), dim3(128), 0, st, A1, lda1, K1, A2, lda2, K2, G, ldg,
                       N, M, workspace);
    RM_CHECK_LAUNCH("rm_dense_wgrad(skinny)");
    const int64_t KNs = (int64_t)K * N;
    const int ngrp = (int)tn_skinny_groups(M);
    float *ws2 = workspace + (int64_t)nblk * KNs;
    hipLaunchKernelGGL(dense_tn_fold_kernel, dim3((unsigned)rm_grid_cap((KNs + 255) / 256, 64), (unsigned)ngrp),
                       dim3(256), 0, st, workspace, nblk, kSkFold, KNs, ws2);
    hipLaunchKernelGGL(dense_tn_reduce, dim3(rm_grid_cap((KNs + 63) / 64, 4096)), dim3(64 * kTrG), 0, st, ws2,
                       ngrp, KNs, N, dW, lddw, accumulate, (const float *)nullptr, (float *)nullptr,
                       (int64_t)0, 0, 0);
    RM_CHECK_LAUNCH("rm_dense_wgrad(reduce)");
    return RM_OK;
  }
  const bool g_fast_rows = ldg % 4 == 0 && rm_aligned16(G) && (N % 4 == 0);
  const TnPlan pl = tn_plan(K, N, M > 0 ? M : 1, g_fast_rows);
  const int kts = pl.kts, nct = pl.nct;
  const int nbt = (N + 31) / 32;
  const int slabs = M > 0 ? pl.slabs : 0;
  float *db_ws = db ? workspace + (int64_t)pl.smax * K * N : nullptr;
  if (M > 0) {
    const int64_t slab_rows = pl.rows_f;
    const int nb0 = nbt < kMaxNB ? nbt : kMaxNB;
    const size_t smem = kTnSmemFloats * sizeof(float);
    // K tiles whose A piece lies inside A1 with 16-byte rows are staged with float4 loads, the
    // ragged last tile per element (block-uniform branch); G likewise when its rows allow it
    const bool g_fast = ldg % 4 == 0 && rm_aligned16(G) && (N % 4 == 0);
    const bool a_al = lda1 % 4 == 0 && rm_aligned16(A1);
    const int kts_fast = a_al ? K1 / kRowsPerBlock : 0;
    RM_REQUIRE((pl.rows_r > slab_rows ? pl.rows_r : slab_rows) * (ldg > lda1 ? ldg : lda1) < ((int64_t)1 << 30),
               "rm_dense_wgrad: rows too long for 32-bit slab offsets");
    TNArgs a{A1, A2, lda1, lda2, K1, K2, G, ldg, N, M, slabs, kts, kts_fast, (a_al && K1 % 4 == 0 && K1 >= 4) ? 1 : 0,
             pl.split_f, pl.live, pl.slabs_r, pl.rows_r, slab_rows, workspace, db_ws};
    const dim3 grid((unsigned)(pl.split_f > 1 ? slabs * (kts - 1) + pl.slabs_r : slabs * kts), (unsigned)nct);
#define RM_TN(P0, P1, FAST_)                                                                      \
  {                                                                                               \
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(dense_tn_kernel<P0, P1, FAST_>),     \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);             \
    hipLaunchKernelGGL((dense_tn_kernel<P0, P1, FAST_>), grid, dim3(kThreads), smem, st, a);      \
  }
#define RM_TN_DISPATCH(FAST_)                \
  if (nb0 <= 2) RM_TN(1, 1, FAST_)           \
  else if (nb0 <= 4) RM_TN(2, 2, FAST_)      \
  else if (nb0 <= 6) RM_TN(3, 3, FAST_)      \
  else if (nb0 <= 8) RM_TN(4, 4, FAST_)      \
  else if (nb0 <= 10) RM_TN(5, 5, FAST_)     \
  else if (nb0 <= 12) RM_TN(6, 6, FAST_)     \
  else if (nb0 == 13) RM_TN(7, 6, FAST_)     \
  else RM_TN(7, 7, FAST_)
    if (g_fast) {
      RM_TN_DISPATCH(true)
    } else {
      RM_TN_DISPATCH(false)
    }
#undef RM_TN_DISPATCH
#undef RM_TN
    RM_CHECK_LAUNCH("rm_dense_wgrad");
  }
  const int64_t KN = (int64_t)K * N;
  hipLaunchKernelGGL(dense_tn_reduce, dim3(rm_grid_cap((KN + N + 63) / 64, 4096)), dim3(64 * kTrG), 0, st,
                     workspace, slabs, KN, N, dW, lddw, accumulate, (const float *)db_ws, db,  // (no slabs: db = 0)
                     pl.split_f > 1 ? (int64_t)(kts - 1) * kRowsPerBlock * N : (int64_t)0,
                     M > 0 ? pl.slabs_r * pl.split_f : 0, slabs);
  RM_CHECK_LAUNCH("rm_dense_wgrad(reduce)");
  return RM_OK;
}
