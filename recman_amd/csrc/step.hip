// DeepFM's whole training step in ONE kernel (rm_deepfm_step): embedding gather, FM, linear term, the skinny MLP,
// PredictionLayer + loss, and EVERY gradient of the step (row gradients d_rows, dlogit, dW0, dW1, biases, output
// projection, the linear term's dense weights) - recman/tf/core/DeepFM.py:107-180 with layers.py:117-128,
// 238-261 (gather), 281-347 (linear), 457-478 (FM), 576-609 (DNN), 796-808 (prediction), utils.py:192-198 (loss).
//
// Why one kernel: every gradient of DeepFM except the parameter reductions is local to the example, so E [B,F,D],
// S, h_l and dh_l never need to reach HBM.  The two-kernel path (rm_embed_mlp_fwd + rm_mlp_bwd) moves 844 MB per
// step at configs[1] (profiles/r02_deepfm_step_bytes.md); this one moves the rows in (one 128-byte line per lookup),
// the row gradients out, and the ids / dense inputs / labels: ~350 MB.
//
// Structure (one 512-thread workgroup per CU, 16-example tiles, v_mfma_f32_16x16x4_f32):
//  * waves 0..6 are WORKERS.  A worker owns up to 4 of the 27 "field slots" (26 embedding fields of 16 k each + the
//    dense inputs as a 16-k pseudo-field), i.e. a k-slice of layer 0: it keeps its slices of W0 in registers in both
//    operand layouts (forward and dX: 2 x 8 registers per slot), keeps its rows of dW0 in accumulators for the whole
//    kernel (8 registers per slot), and ONLY EVER touches its own fields of x - including their part of the FM sums
//    (S, sum of squares) and of the bias / linear-entry sums.  x never crosses waves; what does is small: a worker's
//    partial sums of a tile (3.5 KB: h0, S, squares, bias + linear entries) and the tile's dh0 / g / g*S (5.6 KB).
//  * the rows arrive by LDS-DMA (global_load_lds_dwordx4 with a per-lane source address: one instruction = the 16
//    examples of one field, 64 bytes each), straight into a ring of 3 tile buffers, more than a tile ahead of their
//    use: no register is held by a row in flight.  Lane (example r, piece p) fetches slice p ^ swz(r) of its row,
//    which makes every later ds_read_b128 of "slice q of example n" conflict-free (position q ^ swz(n)) without
//    padding the 64-byte rows.  A field's DMA for tile s + 1 is issued the moment its backward of tile s - 2 has
//    freed the slot - one piece at a time between the matrix work, not 27 pieces at once.
//  * wave 7 is the HEAD: per tile it sums the 7 workers' partial sums, runs layer 1, the output projection, the FM /
//    linear logits, PredictionLayer, the loss term, dLoss/dlogit and the dh chain, publishes dh0 (two layouts), g and
//    g*S for the workers' backward, and accumulates every small gradient (dW1 on the matrix pipe, db0, db1, d w_out,
//    sum g, g^T xd) in registers.  It also turns the ids into row numbers two tiles ahead (the workers' DMA addresses)
//    and stages the dense inputs / labels.
//  * one workgroup barrier per tile.  Segment s (between barriers s-1 and s): workers run backward(s-2) + the DMAs
//    of tile s+1, then forward(s); the head runs epilogue(s-1) meanwhile.  forward(s) -> [barrier] -> epilogue(s) ->
//    [barrier] -> backward(s): the head's serial chain (32 dependent MFMAs + the loss) always overlaps the workers'
//    matrix work of the neighbouring tiles.
//  * per-block partial sums (dW0 slab, small gradients, loss) go to the same finishing launch as rm_mlp_bwd's
//    (mlp_finish_kernel): fixed order, deterministic.
// Measured history and the per-wave phase times: profiles/r03_deepfm_step.md.
#include "mlp_internal.h"
#include "rm_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

constexpr int kWorkers = 7;   // waves 0..6; wave 7 is the head
constexpr int kSlots = 4;     // field slots per worker
constexpr int kMaxF = 26;     // embedding fields (+ the dense pseudo-field = 27 slots over 7 workers: 4 4 4 3 4 4 4)
constexpr int kRing = 3;      // x tile buffers
constexpr int kSlotB = 1024;  // 16 examples x 64 bytes
constexpr int kXBufB = kMaxF * kSlotB;
// one worker's partial sums of a tile: h0 [2 unit halves][64 lanes][4] | S [64][4] | squares [64] | bias/linear sums [64]
constexpr int kPartH = 0, kPartS = 2048, kPartQ = 3072, kPartSd = 3328, kPartB = 3584;
constexpr int kLDT = 36;      // row stride (floats) of the published [example][unit] image of dh0
// published per tile by the head: dh0 in B-operand layout | dh0 as [example][unit] | g*S | g
constexpr int kPubDh = 0, kPubDT = 2048, kPubGS = kPubDT + 16 * kLDT * 4, kPubG = kPubGS + 1024, kPubB = kPubG + 256;
// LDS map (bytes)
constexpr int oX = 0;                                 // [kRing][26][1024] rows of the embedding fields
constexpr int oDense = oX + kRing * kXBufB;           // [4][1024] the dense pseudo-field (written by the head)
constexpr int oJunk = oDense + 4 * kSlotB;            // sink of the DMAs of empty slots (every worker issues 4 per tile)
constexpr int oPart = oJunk + kSlotB;                 // [2][7][kPartB]
constexpr int oPub = oPart + 2 * kWorkers * kPartB;   // [2][kPubB]
constexpr int kRowB = 28 * 64;                        // row numbers of a tile: [28 field slots][16] u32 (the head writes 7 per lane)
constexpr int oRow = oPub + 2 * kPubB;                // [2][kRowB]
constexpr int oHT = oRow + 2 * kRowB;                 // [16][32] h0 as [example][unit]   (head only)
constexpr int oD1T = oHT + 16 * 32 * 4;               // [16][32] dh1 as [example][unit]  (head only)
constexpr int oY = oD1T + 16 * 32 * 4;                // [4][16] labels
constexpr int oW1 = oY + 4 * 16 * 4;                  // [32][32] W1[u][v], 16-byte block c of row u at position c ^ (u & 7)
constexpr int oPar = oW1 + 32 * 32 * 4;               // b0 [32] | b1 [32] | w_out [32] | lin_w_dense [16] | field_off [28] (u32)
constexpr int kParB0 = 0, kParB1 = 128, kParWo = 256, kParWd = 384, kParFo = 448;
constexpr int kLdsBytes = oPar + 448 + 28 * 4;
static_assert(kLdsBytes <= 160 * 1024, "LDS map exceeds a CU");

#ifndef RM_STEP_ABL
#define RM_STEP_ABL 0  // ablation builds (wrong results): 1 no bias / linear entry loads, 2 no d_rows stores, 4 every row = row 0, 8 no DMAs, 16 no worker MFMAs
#endif
// diagnostic build (-DRM_STEP_STAMP, never in the product library): every wave adds up the shader-clock ticks it spends
// in its phases; rm_debug_step_stamps reads them (tools/probe/step_stamps.py).
#ifdef RM_STEP_STAMP
__device__ unsigned long long rm_step_stamp_buf[256 * 8 * 8];
#define ST_NOW() __builtin_amdgcn_s_memtime()
#define ST_DECL unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_t = ST_NOW(), st_t0 = st_t
#define ST_ADD(i) { const unsigned long long st_n = ST_NOW(); st_acc[i] += st_n - st_t; st_t = st_n; }
#define ST_FLUSH(wv) if (lane == 0 && blockIdx.x < 256) { st_acc[7] = ST_NOW() - st_t0; for (int i_ = 0; i_ < 8; ++i_) rm_step_stamp_buf[(blockIdx.x * 8 + (wv)) * 8 + i_] = st_acc[i_]; }
#else
#define ST_DECL
#define ST_ADD(i)
#define ST_FLUSH(wv)
#endif

struct StepArgs {
  const int64_t *idx, *field_off;
  const char *table;
  int64_t row_bytes;
  const float *dense;
  const int64_t *y;
  const float *y_f;
  const float *W0, *b0, *W1, *b1, *w_out, *w0_out, *lin_wd, *lin_w0;
  int F, Dn, H0, H1, act, task, Kp;
  float grad_scale;
  int64_t B;
  float *d_rows, *logit, *pred, *dlogit, *dW0_part, *sg_part, *loss_part;
  // packed output (row-sharded table): the row gradient of occurrence (b, f) goes to row idx[b, f] of d_rows - rows
  // of out_row_bytes = (16 + 4) floats [dE | g | g * lin_mask[f] | 0 0], the send buffer of the backward exchange
  int packed, out_row_bytes, out_rows;
  const float *lin_mask;
};

// activation and its derivative off the post-activation value, branch-free: slope = 0 (relu), 0.2 (leaky_relu,
// tf.nn.leaky_relu's alpha), 1 (identity).  (A runtime `switch (act)` per element compiled to two scalar branches per
// value: ~100 of them per tile in the head wave.)
__device__ __forceinline__ float act_slope(int act) {
  return act == RM_ACT_RELU ? 0.f : (act == RM_ACT_LEAKY_RELU ? 0.2f : 1.f);
}
__device__ __forceinline__ float actf(float v, float slope) { return v > 0.f ? v : slope * v; }
__device__ __forceinline__ float actg(float o, float slope) { return o > 0.f ? 1.f : slope; }

// field slot j of worker w: fields 7 j + w for j < 3; the last round skips worker 3, the head's SIMD partner
// (27 slots over 7 workers: 4 4 4 3 4 4 4).  Slot field == F is the dense pseudo-field.
__device__ __forceinline__ int slot_field(int w, int j) {
  return j < 3 ? 7 * j + w : (w < 3 ? 21 + w : (w > 3 ? 20 + w : -1));
}

// v_mfma_f32_16x16x4_f32: lane (n = lane & 15, q = lane >> 4) supplies A[m = n][k = q] and B[k = q][n]; D[m = 4 q + i][n]
// comes back in register i of lane (n, q).
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16w(float a, float b, f32x4 c) {  // the workers' (ablation: a cheap stand-in)
  if (RM_STEP_ABL & 16) { c.x += a * b; return c; }
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Position of 16-byte slice j of example r's 64-byte row in LDS: j ^ swz(r).  ds_read_b128 serves a wave in the lane
// groups {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32): with lane = (example n = lane & 15, slice q = lane >> 4) a
// group holds every example once, examples 4..11 with the other slice parity - swz = 3 for examples 8..15 gives the
// four rows that share a bank window (n mod 4 equal) four different positions in either group.
__device__ __forceinline__ int swz(int r) { return ((r >> 3) & 1) * 3; }

// One LDS-DMA piece: lane l's 16 bytes at src land at LDS address lds_dst + 16 l (lds_dst wave-uniform).  Through the
// builtin, so that hipcc counts it among the wave's vector-memory operations (its own s_waitcnt for the worker's
// register loads stay exact); hipcc does NOT order LDS reads behind it - that is wait_vm's job.
template <bool NT>
__device__ __forceinline__ void dma16(const char *src, char *lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                   (__attribute__((address_space(3))) void *)lds_dst, 16, 0, NT ? 2 : 0);
}

// waits until at most `younger` (0..24) of this wave's vector-memory operations are outstanding
__device__ __forceinline__ void wait_vm(int younger) {
#define RM_WV(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
  switch (younger) {
    RM_WV(24) RM_WV(23) RM_WV(22) RM_WV(21) RM_WV(20) RM_WV(19) RM_WV(18) RM_WV(17) RM_WV(16) RM_WV(15) RM_WV(14) RM_WV(13) RM_WV(12) RM_WV(11) RM_WV(10) RM_WV(9) RM_WV(8) RM_WV(7) RM_WV(6)
    RM_WV(5) RM_WV(4) RM_WV(3) RM_WV(2) RM_WV(1)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
#undef RM_WV
}

// Sum over the 4 lanes (n, 0..3) of an example = the xor-16 / xor-32 butterfly, on the gfx950 permlane swaps (VALU;
// through ds_bpermute each step is an LDS round trip, and the head wave's epilogue is one dependent chain).
// v_permlane16_swap / v_permlane32_swap exchange the odd rows (upper half) of the first register with the even rows
// (lower half) of the second: with two copies of v the sum of the two results crosses the rows (halves).  Inline asm
// as in cross.hip (hipcc's builtin added the first result to itself); "s_nop 1" = the wait states behind a VALU write.
__device__ __forceinline__ float sum_q(float v) {
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
__device__ __forceinline__ float sum_n(float v) {  // over the 16 lanes (0..15, q)
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---------------------------------------------------------------------------------------------- worker
template <bool NT, bool NT_OUT, bool PACKED>
__device__ __forceinline__ void step_worker(const StepArgs &a, char *smem, const int w, const int lane, const int T) {
  const int n = lane & 15, q = lane >> 4;
  const int F = a.F, H0 = a.H0, K = 16 * F + a.Dn;
  int fld[kSlots];
  bool sv[kSlots], sx[kSlots];  // slot in use; slot is an embedding field (has rows to fetch and a dX)
  float w0f[kSlots][2][4];      // forward A operand: W0[kb + 4 q + ks][16 uh + n]
  float w0x[kSlots][2][4];      // dX A operand:      W0[kb + n][16 uh + 4 q + i]
  f32x4 dw[kSlots][2];          // dW0[kb + 4 q + i][16 uh + n]
#pragma unroll
  for (int j = 0; j < kSlots; ++j) {
    const int f = slot_field(w, j);
    sv[j] = f >= 0 && (f < F || (f == F && a.Dn > 0));
    sx[j] = f >= 0 && f < F;
    fld[j] = sv[j] ? f : 0;
    const int kb = 16 * fld[j];
#pragma unroll
    for (int uh = 0; uh < 2; ++uh) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int kf = kb + 4 * q + e, uf = 16 * uh + n;
        const bool okf = sv[j] && kf < K && uf < H0;
        const float vf = a.W0[okf ? (int64_t)kf * H0 + uf : 0];
        w0f[j][uh][e] = okf ? vf : 0.f;
        const int kx = kb + n, ux = 16 * uh + 4 * q + e;
        const bool okx = sx[j] && kx < K && ux < H0;
        const float vx = a.W0[okx ? (int64_t)kx * H0 + ux : 0];
        w0x[j][uh][e] = okx ? vx : 0.f;
      }
      dw[j][uh] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  const int64_t tstride = gridDim.x;
  // per-lane offsets inside a slot's 1 KiB image, computed once: slice q of example n; element n of examples
  // 4 es + q.  A slot's base is wave-uniform (field, ring buffer): base + lane offset is ONE add per access (a
  // `slot is a field ? ring : dense ring` select on the per-lane pointer was two v_cndmask per access)
  const int lo_slice = n * 64 + 16 * (q ^ swz(n));
  int lo_col[4];
#pragma unroll
  for (int es = 0; es < 4; ++es) lo_col[es] = (4 * es + q) * 64 + 16 * ((n >> 2) ^ swz(4 * es + q)) + 4 * (n & 3);
  const int lo_row = n * F * 64 + 16 * q;  // d_rows: example n, slice q (the field adds 64 f)
  auto slot_base = [&](int j, int tile) {  // LDS byte offset of slot j's image of `tile` (wave-uniform)
    const int ring = oX + ((tile + 6) % 3) * kXBufB + fld[j] * kSlotB, dn = oDense + (tile & 3) * kSlotB;
    return __builtin_amdgcn_readfirstlane((sx[j] || !sv[j]) ? ring : dn);
  };
  // this lane's piece of a row: DMA lane (example lane >> 2, position lane & 3); the bias / linear entry it sums
  // (floats 16 / 17 of the row: lanes with (lane & 1) == 0 take the bias entry, the others the linear weight)
  const int piece = 16 * ((lane & 3) ^ swz(lane >> 2));
  const int side_off = 64 + 4 * (lane & 1);
  float sdv[kSlots];  // bias / linear entries of the tile whose rows are in flight
#pragma unroll
  for (int j = 0; j < kSlots; ++j) sdv[j] = 0.f;
  // PACKED: the row numbers (= positions in the exchange buffers) of example n's occurrences, kept from the segment
  // that requests a tile's rows (tile s + 1 in segment s) to the one that stores its gradients (tile s - 2)
  unsigned rp[4][kSlots];
  float lmask[kSlots];  // linear_features subset: the field's factor on the linear-entry gradient
#pragma unroll
  for (int j = 0; j < kSlots; ++j) {
    rp[0][j] = rp[1][j] = rp[2][j] = rp[3][j] = 0;
    lmask[j] = (PACKED && a.lin_mask != nullptr && sx[j]) ? a.lin_mask[fld[j]] : 1.f;
  }
  ST_DECL;
  ST_ADD(0);  // prologue

  for (int s = -2; s <= T + 1; ++s) {
    const bool hasB = s - 2 >= 0 && s - 2 < T;  // backward of tile s - 2 (4 d_rows stores)
    const bool hasD = s + 1 >= 0 && s + 1 < T;  // rows of tile s + 1 (4 DMAs + 4 entry loads)
    // ---- the bias / linear entries of tile s (loaded a segment ago), before their registers are loaded again
    float side_sum = 0.f;
#pragma unroll
    for (int j = 0; j < kSlots; ++j) side_sum += sx[j] ? sdv[j] : 0.f;
    // ------------------------------------------------------------ backward of tile s - 2, rows of tile s + 1
    // Phases WITHOUT per-slot branches (one basic block each, so that hipcc interleaves the slots' MFMA chains and
    // keeps their LDS reads in flight together): R read everything the backward needs of the 4 slots, D issue the
    // DMAs / entry loads of tile s + 1 into the slots just read, M 64 MFMAs, S FM term + stores.  Empty slots run on
    // field 0's data with zero weights (their dW0 accumulators are never stored, their d_rows store is dropped).
    {
      const int tb = s - 2, td = s + 1;
      const char *pub = smem + oPub + (tb & 1) * kPubB;
      char *xd = smem + oX + ((td + 3) % 3) * kXBufB;  // (the same buffer: tile s + 1 replaces tile s - 2)
      f32x4 dhB[2], gS, e4[kSlots];
      float dT[2][4], g, col[kSlots][4];
      if (hasB) {  // ---- R
#pragma unroll
        for (int j = 0; j < kSlots; ++j) {
          const char *xs = smem + slot_base(j, tb);
#pragma unroll
          for (int es = 0; es < 4; ++es)  // x[example 4 es + q][k = n]: slice n >> 2 of that row
            col[j][es] = *reinterpret_cast<const float *>(xs + lo_col[es]);
          e4[j] = *reinterpret_cast<const f32x4 *>(xs + lo_slice);
        }
#pragma unroll
        for (int uh = 0; uh < 2; ++uh) {
          dhB[uh] = *reinterpret_cast<const f32x4 *>(pub + kPubDh + (uh * 64 + lane) * 16);
#pragma unroll
          for (int es = 0; es < 4; ++es)
            dT[uh][es] = *reinterpret_cast<const float *>(pub + kPubDT + ((4 * es + q) * kLDT + 16 * uh + n) * 4);
        }
        gS = *reinterpret_cast<const f32x4 *>(pub + kPubGS + lane * 16);
        g = *reinterpret_cast<const float *>(pub + kPubG + lane * 4);
      }
      ST_ADD(1);  // R
      // ---- D, M, S interleaved: a vector-memory instruction waits for a slot of the CU's memory pipeline (the
      // younger waves of a SIMD pair sat up to 6,000 cycles per tile in a burst of 8 of them, tools/probe/
      // step_stamps.py) - issued one at a time between groups of 16 MFMAs, the wait runs under the matrix work
      unsigned rid[kSlots];
      if (PACKED) {  // after this: rp[g] = tile s + 1 - g; rp[3] = tile s - 2, whose gradients are stored below
#pragma unroll
        for (int j = 0; j < kSlots; ++j) {
          rp[3][j] = rp[2][j];
          rp[2][j] = rp[1][j];
          rp[1][j] = rp[0][j];
        }
      }
      if (hasD) {
        const unsigned *rowid = reinterpret_cast<const unsigned *>(smem + oRow + (td & 1) * kRowB);
#pragma unroll
        for (int j = 0; j < kSlots; ++j) {
          rid[j] = rowid[fld[j] * 16 + (lane >> 2)];
          if (PACKED) rp[0][j] = rowid[fld[j] * 16 + n];
        }
      }
      // the old rows have been READ (their values are in registers) before the new ones are requested
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const int64_t ex0 = ((int64_t)blockIdx.x + tb * tstride) * 16;
      const int64_t left = a.B - ex0;
      const int rows_t = left < 16 ? (int)left : 16;
      const rsrc_t rd = PACKED ? __builtin_amdgcn_make_buffer_rsrc(a.d_rows, 0, hasB ? a.out_rows * a.out_row_bytes : 0, 0x00020000)
                               : __builtin_amdgcn_make_buffer_rsrc(a.d_rows + (hasB ? ex0 * F * 16 : 0), 0,
                                                                   hasB ? rows_t * F * 64 : 0, 0x00020000);
      const bool ex_ok = n < rows_t;
      // the row gradient of slot i (slots without an embedding field and examples past B: out of range, the store is
      // dropped - every tile issues the same number of them).  PACKED: to row rp[3][i] of the exchange buffer, and
      // the lanes of slice 0 add the row's tail [g_bias | g_lin | 0 0]
      auto store_row = [&](int i, const f32x4 &o) {
        const bool ok = sx[i] && !(RM_STEP_ABL & 2);
        if (PACKED) {
          const int base = (ok && ex_ok) ? (int)rp[3][i] * a.out_row_bytes : 0x7ffffff0 - 64;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rd, base + 16 * q, 0, NT_OUT ? 2 : 0);
          const f32x4 tail = f32x4{g, g * lmask[i], 0.f, 0.f};
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, tail), rd, q == 0 ? base + 64 : 0x7ffffff0,
                                                 0, NT_OUT ? 2 : 0);
        } else {
          const int off = lo_row + __builtin_amdgcn_readfirstlane(ok ? fld[i] * 64 : 0x70000000);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rd, off, 0, NT_OUT ? 2 : 0);
        }
      };
      f32x4 acc[kSlots];
#pragma unroll
      for (int j = 0; j < kSlots; ++j) {
        if (hasD) {
          // empty slots fetch row 0 into the sink: every worker issues the same operations per tile, so the
          // hand-counted wait of the forward holds for all of them
          const char *row = a.table + ((RM_STEP_ABL & 4) || !sx[j] ? 0 : (int64_t)rid[j] * a.row_bytes);
          if (!(RM_STEP_ABL & 8)) dma16<NT>(row + piece, sx[j] ? xd + fld[j] * kSlotB : smem + oJunk);
          if (!(RM_STEP_ABL & 1)) sdv[j] = *reinterpret_cast<const float *>(row + side_off);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (hasB) {
          acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int es = 0; es < 4; ++es) {
            dw[j][0] = mfma16w(col[j][es], dT[0][es], dw[j][0]);
            dw[j][1] = mfma16w(col[j][es], dT[1][es], dw[j][1]);
            acc[j] = mfma16w(w0x[j][0][es], dhB[0][es], acc[j]);
            acc[j] = mfma16w(w0x[j][1][es], dhB[1][es], acc[j]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (hasB && j > 0) {
          // the PREVIOUS slot's row gradient: dLoss/dE = dX + g (S - E)   (FM second order, layers.py:468-476);
          // slots without an embedding field: out of range, the store is dropped - every tile issues 4 of them
          const int i = j - 1;
          f32x4 o;
          o.x = acc[i].x + (gS.x - g * e4[i].x);
          o.y = acc[i].y + (gS.y - g * e4[i].y);
          o.z = acc[i].z + (gS.z - g * e4[i].z);
          o.w = acc[i].w + (gS.w - g * e4[i].w);
          store_row(i, o);
        }
      }
      if (hasB) {
        const int i = kSlots - 1;
        f32x4 o;
        o.x = acc[i].x + (gS.x - g * e4[i].x);
        o.y = acc[i].y + (gS.y - g * e4[i].y);
        o.z = acc[i].z + (gS.z - g * e4[i].z);
        o.w = acc[i].w + (gS.w - g * e4[i].w);
        store_row(i, o);
      }
    }
    ST_ADD(6);  // M + S
    // ------------------------------------------------------------ forward of tile s
    if (s >= 0 && s < T) {
      // the tile's last DMA (slot 3, segment s - 1) is followed by that slot's entry load and the segment's last two
      // stores (slots 2 and 3), then by this segment's operations
      constexpr int SP = PACKED ? 2 : 1;  // stores per slot
      const int younger = 1 + ((s - 3 >= 0) ? 2 * SP : 0) + 8 * (hasD ? 1 : 0) + 4 * SP * (hasB ? 1 : 0);
      wait_vm(younger);
      ST_ADD(3);  // wait for the tile's rows
      f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0, S = acc0, Q = acc0;
      f32x4 x4[kSlots];
#pragma unroll
      for (int j = 0; j < kSlots; ++j) {
        x4[j] = *reinterpret_cast<const f32x4 *>(smem + slot_base(j, s) + lo_slice);
      }
#pragma unroll
      for (int j = 0; j < kSlots; ++j) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          acc0 = mfma16w(w0f[j][0][ks], x4[j][ks], acc0);
          acc1 = mfma16w(w0f[j][1][ks], x4[j][ks], acc1);
        }
        // FM sums of this worker's fields (layers.py:467-476): slice q of example n
        const float m = sx[j] ? 1.f : 0.f;
        S += m * x4[j];
        Q += m * (x4[j] * x4[j]);
      }
      char *part = smem + oPart + ((s & 1) * kWorkers + w) * kPartB;
      *reinterpret_cast<f32x4 *>(part + kPartH + lane * 16) = acc0;
      *reinterpret_cast<f32x4 *>(part + kPartH + (64 + lane) * 16) = acc1;
      *reinterpret_cast<f32x4 *>(part + kPartS + lane * 16) = S;
      *reinterpret_cast<float *>(part + kPartQ + lane * 4) = (Q.x + Q.y) + (Q.z + Q.w);
      *reinterpret_cast<float *>(part + kPartSd + lane * 4) = side_sum;
    }
    ST_ADD(4);  // forward
    __syncthreads();
    ST_ADD(5);  // barrier
  }
  // ---- this block's dW0 slab: every worker owns the rows of its fields
#pragma unroll
  for (int j = 0; j < kSlots; ++j)
    if (sv[j]) {
#pragma unroll
      for (int uh = 0; uh < 2; ++uh)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          a.dW0_part[((int64_t)blockIdx.x * a.Kp + 16 * fld[j] + 4 * q + i) * 32 + 16 * uh + n] = dw[j][uh][i];
    }
  ST_ADD(0);
  ST_FLUSH(w);
}

// ---------------------------------------------------------------------------------------------- head
__device__ __forceinline__ void step_head(const StepArgs &a, char *smem, const int lane, const int T) {
  const int n = lane & 15, q = lane >> 4;
  const int F = a.F, Dn = a.Dn, H0 = a.H0, H1 = a.H1;
  const float act = act_slope(a.act);
  const int64_t B = a.B, tstride = gridDim.x;
  // ---- parameters: in LDS, read per tile (as registers they pushed this wave past 256 VGPRs: W1 in both operand
  // layouts alone is 32).  W1[u][v] row-major with the 16-byte blocks of a row XOR-swizzled by the row, so that the
  // forward's column reads (ds_read_b32, lanes along v) and the chain's row reads (ds_read_b128) both spread over the banks.
  {
    float *w1s = reinterpret_cast<float *>(smem + oW1);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int e = lane + 64 * i, u = e >> 5, v = e & 31;
      const bool ok = u < H0 && v < H1;
      const float f = a.W1[ok ? u * H1 + v : 0];
      w1s[u * 32 + 4 * ((v >> 2) ^ (u & 7)) + (v & 3)] = ok ? f : 0.f;
    }
    float *par = reinterpret_cast<float *>(smem + oPar);
    if (lane < 32) {
      const float f0 = a.b0[lane < H0 ? lane : 0], f1 = a.b1[lane < H1 ? lane : 0], f2 = a.w_out[lane < H1 ? lane : 0];
      par[kParB0 / 4 + lane] = lane < H0 ? f0 : 0.f;
      par[kParB1 / 4 + lane] = lane < H1 ? f1 : 0.f;
      par[kParWo / 4 + lane] = lane < H1 ? f2 : 0.f;
      const float f3 = a.lin_wd[lane < Dn ? lane : 0];
      if (lane < 16) par[kParWd / 4 + lane] = lane < Dn ? f3 : 0.f;
      if (lane < 28) reinterpret_cast<unsigned *>(par)[kParFo / 4 + lane] = (unsigned)a.field_off[lane < F ? lane : F - 1];
    }
  }
  const float *w1s = reinterpret_cast<const float *>(smem + oW1);
  const char *par = smem + oPar;
  const float w0o = a.w0_out[0], lw0 = a.lin_w0[0];

  // ---- accumulators of the small gradients
  f32x4 dW1[2][2];
  float db0[2][4], db1[2][4], dwo[2][4], dxd[4], sg = 0.f, loss_acc = 0.f;
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      dW1[x][0][i] = dW1[x][1][i] = 0.f;
      db0[x][i] = db1[x][i] = dwo[x][i] = 0.f;
    }
#pragma unroll
  for (int i = 0; i < 4; ++i) dxd[i] = 0.f;

  auto ex_of = [&](int t) {  // this lane's example of tile t, clamped into the batch
    const int64_t b = ((int64_t)blockIdx.x + t * tstride) * 16 + n;
    return b < B ? b : B - 1;
  };
  // prefetch registers.  Everything about this lane's fields q + 4 j is UNCONDITIONAL: fields past F read field
  // F - 1 again and land in row-number slots nobody reads (per-lane branches around the loads made hipcc serialise
  // them behind s_waitcnt vmcnt(0): 7 exposed HBM round trips per tile).  Loaded values are used RAW a segment
  // later: a select or conversion next to a load waits for it - and for every load issued before it.
  unsigned idr[7];  // ids (low words: row numbers < 2^32) of tile s + 2, loaded one segment earlier
  float dnr[4];     // dense columns 4 q .. 4 q + 3 of tile s + 1
  unsigned yr = 0;  // ... and its label (low word of the int64 / the float's bits)
#pragma unroll
  for (int j = 0; j < 7; ++j) idr[j] = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) dnr[i] = 0.f;
  auto load_ids = [&](int t) {
    const unsigned *p = reinterpret_cast<const unsigned *>(a.idx + ex_of(t) * F);
#pragma unroll
    for (int j = 0; j < 7; ++j) idr[j] = p[2 * (q + 4 * j < F ? q + 4 * j : F - 1)];
  };
  if (T > 0) load_ids(0);
  ST_DECL;
  ST_ADD(0);

  for (int s = -2; s <= T + 1; ++s) {
    // ------------------------------------------------------------ epilogue of tile s - 1
    if (s - 1 >= 0 && s - 1 < T) {
      const int t = s - 1;
      const int64_t bex = ((int64_t)blockIdx.x + t * tstride) * 16 + n;
      const bool valid = bex < B;
      // the 7 workers' partial sums, fixed order: h0 (pre-activation), FM sums, bias / linear entries
      f32x4 pre[2], S = f32x4{0.f, 0.f, 0.f, 0.f};
      pre[0] = pre[1] = S;
      float ss = 0.f, y1 = 0.f, ls = 0.f;
#pragma unroll
      for (int w = 0; w < kWorkers; ++w) {
        const char *part = smem + oPart + ((t & 1) * kWorkers + w) * kPartB;
        pre[0] += *reinterpret_cast<const f32x4 *>(part + kPartH + lane * 16);
        pre[1] += *reinterpret_cast<const f32x4 *>(part + kPartH + (64 + lane) * 16);
        S += *reinterpret_cast<const f32x4 *>(part + kPartS + lane * 16);
        ss += *reinterpret_cast<const float *>(part + kPartQ + lane * 4);
        const float2 sd = *reinterpret_cast<const float2 *>(part + kPartSd + 16 * n);  // DMA lanes 4 n, 4 n + 1
        y1 += sd.x;
        ls += sd.y;
        if (w == 3) __builtin_amdgcn_sched_barrier(0);  // (two batches of reads)
      }
      const f32x4 dn4 = Dn > 0 ? *reinterpret_cast<const f32x4 *>(smem + oDense + (t & 3) * kSlotB + n * 64 + 16 * (q ^ swz(n)))
                               : f32x4{0.f, 0.f, 0.f, 0.f};
      // FM and linear logits (their reductions ahead of the layer-1 MFMAs)
      const f32x4 wdv = *reinterpret_cast<const f32x4 *>(par + kParWd + 16 * q);
      const float fmq = sum_q(S.x * S.x + S.y * S.y + S.z * S.z + S.w * S.w - ss);
      const float lind = sum_q(dn4.x * wdv.x + dn4.y * wdv.y + dn4.z * wdv.z + dn4.w * wdv.w);
      const float fm = y1 + 0.5f * fmq;
      const float lin = ls + lind + lw0;
      const float ty = *reinterpret_cast<const float *>(smem + oY + ((t & 3) * 16 + n) * 4);
      ST_ADD(1);  // partial sums
      // layer 0 activation; h0 as [example][unit] for dW1
      float h0[2][4], h1[2][4];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const f32x4 b0v = *reinterpret_cast<const f32x4 *>(par + kParB0 + (16 * h + 4 * q) * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) h0[h][i] = actf(pre[h][i] + b0v[i], act);
        *reinterpret_cast<f32x4 *>(smem + oHT + (n * 32 + 16 * h + 4 * q) * 4) =
            f32x4{h0[h][0], h0[h][1], h0[h][2], h0[h][3]};
      }
      // layer 1
      f32x4 acc[2];
      acc[0] = acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int uh = 0; uh < 2; ++uh)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int u = 16 * uh + 4 * q + i;  // A operand of k-step (uh, i): W1[u][16 vh + n]
          const float a0 = w1s[u * 32 + 4 * ((n >> 2) ^ (u & 7)) + (n & 3)];
          const float a1 = w1s[u * 32 + 4 * ((4 + (n >> 2)) ^ (u & 7)) + (n & 3)];
          acc[0] = mfma16(a0, h0[uh][i], acc[0]);
          acc[1] = mfma16(a1, h0[uh][i], acc[1]);
        }
      float dnn = 0.f;
      f32x4 wov[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const f32x4 b1v = *reinterpret_cast<const f32x4 *>(par + kParB1 + (16 * h + 4 * q) * 4);
        wov[h] = *reinterpret_cast<const f32x4 *>(par + kParWo + (16 * h + 4 * q) * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          h1[h][i] = actf(acc[h][i] + b1v[i], act);
          dnn += h1[h][i] * wov[h][i];
        }
      }
      dnn = sum_q(dnn) + w0o;
      // PredictionLayer + loss (rm_logit_loss's arithmetic, same order of the branch sum)
      float z = 0.f;
      z += lin;
      z += fm;
      z += dnn;
      float p, dz;
      const float lt = rm_loss_point(z, ty, a.task, &p, &dz);
      float gb = dz * (1.0f / (float)B);
      gb *= a.grad_scale;
      gb = valid ? gb : 0.f;
      if (valid && q == 0) {
        a.logit[bex] = z;
        a.pred[bex] = p;
        a.dlogit[bex] = gb;
        loss_acc += lt;
      }
      ST_ADD(2);  // layer 1, logits, loss
      // dh chain
      float dh1[2][4], dh0[2][4];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int i = 0; i < 4; ++i) dh1[h][i] = gb * wov[h][i] * actg(h1[h][i], act);
        *reinterpret_cast<f32x4 *>(smem + oD1T + (n * 32 + 16 * h + 4 * q) * 4) =
            f32x4{dh1[h][0], dh1[h][1], dh1[h][2], dh1[h][3]};
      }
      acc[0] = acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int vh = 0; vh < 2; ++vh) {
        // A operands of k-steps (vh, 0..3): W1[16 uh + n][16 vh + 4 q + i], one 16-byte block per output half
        const f32x4 c0 = *reinterpret_cast<const f32x4 *>(w1s + n * 32 + 4 * ((4 * vh + q) ^ (n & 7)));
        const f32x4 c1 = *reinterpret_cast<const f32x4 *>(w1s + (16 + n) * 32 + 4 * ((4 * vh + q) ^ (n & 7)));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc[0] = mfma16(c0[i], dh1[vh][i], acc[0]);
          acc[1] = mfma16(c1[i], dh1[vh][i], acc[1]);
        }
      }
      char *pub = smem + oPub + (t & 1) * kPubB;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int i = 0; i < 4; ++i) dh0[h][i] = acc[h][i] * actg(h0[h][i], act);
        const f32x4 d4 = f32x4{dh0[h][0], dh0[h][1], dh0[h][2], dh0[h][3]};
        *reinterpret_cast<f32x4 *>(pub + kPubDh + (h * 64 + lane) * 16) = d4;
        *reinterpret_cast<f32x4 *>(pub + kPubDT + (n * kLDT + 16 * h + 4 * q) * 4) = d4;
      }
      *reinterpret_cast<f32x4 *>(pub + kPubGS + lane * 16) = f32x4{gb * S.x, gb * S.y, gb * S.z, gb * S.w};
      *reinterpret_cast<float *>(pub + kPubG + lane * 4) = gb;
      ST_ADD(3);  // chain + publish
      // small gradients: dW1 += h0^T dh1 on the matrix pipe, the rest per lane
#pragma unroll
      for (int es = 0; es < 4; ++es) {
        float av[2], bv[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          av[h] = *reinterpret_cast<const float *>(smem + oHT + ((4 * es + q) * 32 + 16 * h + n) * 4);
          bv[h] = *reinterpret_cast<const float *>(smem + oD1T + ((4 * es + q) * 32 + 16 * h + n) * 4);
        }
#pragma unroll
        for (int uh = 0; uh < 2; ++uh)
#pragma unroll
          for (int vh = 0; vh < 2; ++vh) dW1[uh][vh] = mfma16(av[uh], bv[vh], dW1[uh][vh]);
      }
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          db0[h][i] += dh0[h][i];
          db1[h][i] += dh1[h][i];
          dwo[h][i] += gb * h1[h][i];
        }
      sg += q == 0 ? gb : 0.f;
      dxd[0] += gb * dn4.x;
      dxd[1] += gb * dn4.y;
      dxd[2] += gb * dn4.z;
      dxd[3] += gb * dn4.w;
    }
    ST_ADD(4);  // small gradients
    // ------------------------------------------------------------ the prefetch chain: FIRST everything that consumes
    // registers loaded one segment ago, THEN this segment's loads (a conservative s_waitcnt vmcnt(0) in front of a
    // consumer must not find loads that have only just been issued)
    if (s + 1 >= 0 && s + 1 < T) {  // dense inputs / label of tile s + 1 -> LDS (columns past Dn masked here)
      const int t = s + 1;
      if (Dn > 0)
        *reinterpret_cast<f32x4 *>(smem + oDense + (t & 3) * kSlotB + n * 64 + 16 * (q ^ swz(n))) =
            f32x4{4 * q + 0 < Dn ? dnr[0] : 0.f, 4 * q + 1 < Dn ? dnr[1] : 0.f, 4 * q + 2 < Dn ? dnr[2] : 0.f,
                  4 * q + 3 < Dn ? dnr[3] : 0.f};
      if (q == 0)
        *reinterpret_cast<float *>(smem + oY + ((t & 3) * 16 + n) * 4) = a.y ? (float)(int)yr : __uint_as_float(yr);
    }
    if (s + 2 < T) {  // row numbers of tile s + 2 -> LDS (slots >= F: never read)
      unsigned *rowid = reinterpret_cast<unsigned *>(smem + oRow + ((s + 2) & 1) * kRowB);
#pragma unroll
      for (int j = 0; j < 7; ++j)
        rowid[(q + 4 * j) * 16 + n] = idr[j] + *reinterpret_cast<const unsigned *>(par + kParFo + (q + 4 * j) * 4);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (s + 2 >= 0 && s + 2 < T) {
      const int64_t b = ex_of(s + 2);
      if (Dn > 0) {
        const float *dp = a.dense + b * Dn;
#pragma unroll
        for (int i = 0; i < 4; ++i) dnr[i] = dp[4 * q + i < Dn ? 4 * q + i : 0];
      }
      const unsigned *yp = a.y ? reinterpret_cast<const unsigned *>(a.y + b) : reinterpret_cast<const unsigned *>(a.y_f + b);
      yr = *yp;
    }
    if (s + 3 < T) load_ids(s + 3);
    ST_ADD(6);  // prefetch chain
    __syncthreads();
    ST_ADD(5);  // barrier
  }

  // ---- this block's partial of the small gradients and the loss
  float *sgp = a.sg_part + (int64_t)blockIdx.x * kRmSgStride;
#pragma unroll
  for (int uh = 0; uh < 2; ++uh)
#pragma unroll
    for (int vh = 0; vh < 2; ++vh)
#pragma unroll
      for (int i = 0; i < 4; ++i) sgp[(16 * uh + 4 * q + i) * 32 + 16 * vh + n] = dW1[uh][vh][i];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float s0 = sum_n(db0[h][i]), s1 = sum_n(db1[h][i]), s2 = sum_n(dwo[h][i]);
      if (n == 0) {
        const int u = 16 * h + 4 * q + i;
        sgp[2 * 1024 + u] = s0;
        sgp[2 * 1024 + 32 + u] = s1;
        sgp[2 * 1024 + 64 + u] = s2;  // NL = 2: d w_out behind the two db_l
      }
    }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float sx = sum_n(dxd[i]);
    if (n == 0) sgp[kRmSgDense + 4 * q + i] = sx;
  }
  const float sgt = rm_wave_sum(sg), lst = rm_wave_sum(loss_acc);
  if (lane == 0) {
    sgp[2 * 1024 + 64 + 32] = sgt;
    a.loss_part[blockIdx.x] = lst;
  }
  ST_ADD(0);
  ST_FLUSH(7);
}

template <bool NT, bool NT_OUT, bool PACKED>
__global__ __launch_bounds__(512) void deepfm_step_kernel(StepArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t ntiles = (a.B + 15) / 16;
  const int T = (int)((ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x);  // tiles blockIdx.x + i gridDim.x
#ifndef RM_STEP_PRIO
#define RM_STEP_PRIO 1
#endif
  if (wave < kWorkers) {
    // static priority for the younger half: waves 4..6 lose every arbitration (issue ports, the vector-memory queue)
    // against their SIMD partners 0..2 otherwise and run their phases AFTER them instead of beside them
    if (RM_STEP_PRIO == 1 && wave >= 4) __builtin_amdgcn_s_setprio(1);
    step_worker<NT, NT_OUT, PACKED>(a, smem, wave, lane, T);
  } else {
    if (RM_STEP_PRIO) __builtin_amdgcn_s_setprio(2);  // the head's epilogue is one dependent chain: never make it queue
    step_head(a, smem, lane, T);
  }
}

}  // namespace

#ifdef RM_STEP_STAMP
extern "C" int rm_debug_step_stamps(unsigned long long *host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(rm_step_stamp_buf), sizeof(unsigned long long) * n);
}
#endif

extern "C" int64_t rm_deepfm_step_workspace(int F, int Dn) {
  const int Kp = ((16 * F + Dn + 63) / 64) * 64;
  return (int64_t)256 * Kp * 32 + (int64_t)256 * kRmSgStride + 256;
}

extern "C" int rm_deepfm_step_supported(int F, int D, int64_t table_ld, int Dn, int NL, const int *H) {
  if (D != 16 || table_ld < 20 || table_ld % 4 != 0 || F < 1 || F > 26 || Dn < 0 || Dn > 16 || NL != 2 || !H) return 0;
  return H[0] >= 1 && H[0] <= 32 && H[1] >= 1 && H[1] <= 32;
}

extern "C" int rm_deepfm_step(const int64_t *idx, const float *table, int64_t table_ld, const int64_t *field_off,
                              const float *dense, int Dn, const int64_t *y, const float *y_f, int64_t B, int F,
                              int D, int NL, const int *H, const float *const *W, const float *const *bias,
                              const float *w_out, const float *w0_out, const float *lin_w_dense,
                              const float *lin_w0, int act, int task, float grad_scale, float *d_rows,
                              float *logit, float *pred, float *dlogit, float *loss, float *const *dW,
                              float *const *db, float *d_w_out, float *d_w0_out, float *d_lin_w_dense,
                              float *d_lin_w0, float *workspace, int64_t packed_rows, const float *lin_field_mask,
                              int flags, rm_stream_t stream) {
  RM_REQUIRE(H && rm_deepfm_step_supported(F, D, table_ld, Dn, NL, H),
             "rm_deepfm_step: needs D = 16, fused rows (table_ld >= 20, a multiple of 4), F <= 26, Dn <= 16 and two "
             "hidden layers of width <= 32");
  RM_REQUIRE(B >= 0, "rm_deepfm_step: B < 0");
  if (B == 0) return RM_OK;
  RM_REQUIRE(idx && table && field_off && (Dn == 0 || (dense && lin_w_dense)) && W && bias && W[0] && W[1] &&
                 bias[0] && bias[1] && w_out && w0_out && lin_w0 && d_rows && logit && pred && dlogit && loss && dW &&
                 dW[0] && dW[1] && workspace,
             "rm_deepfm_step: NULL argument");
  RM_REQUIRE((y != nullptr) != (y_f != nullptr), "rm_deepfm_step: exactly one of y / y_f");
  RM_REQUIRE(task == 0 || task == 1, "rm_deepfm_step: task must be 0 or 1");
  RM_REQUIRE(rm_aligned16(table) && rm_aligned16(d_rows), "rm_deepfm_step: table / d_rows must be 16-byte aligned");
  RM_REQUIRE(packed_rows >= 0 && packed_rows * 80 < (1ll << 31), "rm_deepfm_step: packed_rows out of range");
  const int K = 16 * F + Dn, Kp = ((K + 63) / 64) * 64;
  const int64_t ntiles = (B + 15) / 16;
  const int nblk = rm_grid_cap(ntiles, 256);  // one 8-wave workgroup per CU
  StepArgs a;
  a.idx = idx; a.field_off = field_off; a.table = reinterpret_cast<const char *>(table); a.row_bytes = table_ld * 4;
  a.dense = dense; a.y = y; a.y_f = y_f;
  a.W0 = W[0]; a.b0 = bias[0]; a.W1 = W[1]; a.b1 = bias[1]; a.w_out = w_out; a.w0_out = w0_out;
  a.lin_wd = Dn > 0 ? lin_w_dense : w_out; a.lin_w0 = lin_w0;
  a.F = F; a.Dn = Dn; a.H0 = H[0]; a.H1 = H[1]; a.act = act; a.task = task; a.Kp = Kp;
  a.grad_scale = grad_scale; a.B = B;
  a.d_rows = d_rows; a.logit = logit; a.pred = pred; a.dlogit = dlogit;
  a.packed = packed_rows > 0; a.out_row_bytes = 80; a.out_rows = (int)packed_rows; a.lin_mask = lin_field_mask;
  a.dW0_part = workspace;
  a.sg_part = workspace + (int64_t)256 * Kp * 32;
  a.loss_part = a.sg_part + (int64_t)256 * kRmSgStride;
  hipStream_t st = (hipStream_t)stream;
  const bool nt = (flags & 1) != 0, nt_out = (flags & 2) != 0;
#define RM_STEP(NT_, NTO_, PK_)                                                                              \
  {                                                                                                          \
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(deepfm_step_kernel<NT_, NTO_, PK_>),            \
                              hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);                        \
    hipLaunchKernelGGL((deepfm_step_kernel<NT_, NTO_, PK_>), dim3(nblk), dim3(512), kLdsBytes, st, a);       \
  }
  if (a.packed) {
    if (nt) RM_STEP(true, false, true) else RM_STEP(false, false, true)
  } else if (nt) {
    if (nt_out) RM_STEP(true, true, false) else RM_STEP(true, false, false)
  } else {
    if (nt_out) RM_STEP(false, true, false) else RM_STEP(false, false, false)
  }
#undef RM_STEP
  RM_CHECK_LAUNCH("rm_deepfm_step");
  if (flags & 4) return RM_OK;  // (measurement: the step kernel alone, the per-block partial sums stay in the workspace)
  return rm_internal_mlp_finish(a.dW0_part, nblk, K, Kp, H[0], dW[0], a.sg_part, nblk, 2, H, dW, db, d_w_out,
                                d_w0_out, Dn > 0 ? d_lin_w_dense : nullptr, d_lin_w0, Dn, a.loss_part, nblk, B, loss,
                                st);
}
