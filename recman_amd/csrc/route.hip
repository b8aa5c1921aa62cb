// Row-sharded table routing (multi-GPU; new - the reference is single-device).
// Global row r = field_off[f] + idx[b,f] lives on rank r % W at local row r / W.
// rm_shard_route buckets the n = B*F occurrences by owner with a deterministic counting
// sort (count per block -> scan -> stable in-block ranking):
//   pos[o]            position of occurrence o in the owner-bucketed order
//   send_ids[pos[o]]  its local row on the owner
//   counts[w]         occurrences owned by rank w
// rm_pack_grad_rows writes the per-occurrence gradient rows [dE | g_bias | g_lin | 0..]
// straight into bucketed order (the send buffer of the backward all_to_all).
#include "rm_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kMaxW = 16;
constexpr int kMaxB = kMaxW + 1;  // buckets: one per owner + one BEHIND them for the empty occurrences (idx < 0:
                                  // the padding of a multi-valued feature's tag columns) - they get pos = -1 and
                                  // take no slot

__global__ __launch_bounds__(kBlock) void route_count_kernel(
    const int64_t *__restrict__ idx, const int64_t *__restrict__ field_off, int64_t n, int F, int W,
    int64_t per_block, int *__restrict__ cnt /* [W][nblk] */, int64_t *__restrict__ fill,
    int64_t nfill) {
  // fixed-capacity layout: every slot starts as the empty id -1 (the place kernel, next on the
  // stream, overwrites the occupied ones).  A kernel, not hipMemsetAsync: the step is captured in
  // hipGraphs and a kernel node is the one node type every kernel of this library already is.
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < nfill; i += (int64_t)gridDim.x * kBlock)
    fill[i] = -1;
  __shared__ int sc[kMaxB];
  if (threadIdx.x < kMaxB) sc[threadIdx.x] = 0;
  __syncthreads();
  const int64_t o0 = (int64_t)blockIdx.x * per_block;
  const int64_t o1 = o0 + per_block < n ? o0 + per_block : n;
  int local[kMaxB];
#pragma unroll
  for (int w = 0; w < kMaxB; ++w) local[w] = 0;
  for (int64_t o = o0 + threadIdx.x; o < o1; o += kBlock) {
    const int64_t id = idx[o];
    const int w = id < 0 ? W : (int)((field_off[o % F] + id) % W);
#pragma unroll
    for (int q = 0; q < kMaxB; ++q) local[q] += (q == w);
  }
#pragma unroll
  for (int w = 0; w < kMaxB; ++w) {
    if (w <= W) {
      const int s = (int)rm_wave_sum((float)local[w]);  // exact: counts < 2^24
      if ((threadIdx.x & 63) == 0 && s) atomicAdd(&sc[w], s);
    }
  }
  __syncthreads();
  if ((int)threadIdx.x <= W) cnt[threadIdx.x * gridDim.x + blockIdx.x] = sc[threadIdx.x];
}

// exclusive scan over the w-major (w, block) grid; one block, sequential chunks
__global__ __launch_bounds__(1024) void route_scan_kernel(int *__restrict__ cnt, int total,
                                                          int nblk, int W,
                                                          int64_t *__restrict__ counts) {
  __shared__ int part[1024];
  const int tid = threadIdx.x;
  const int per = (total + 1023) / 1024;
  int s = 0;
  for (int i = tid * per; i < (tid + 1) * per && i < total; ++i) s += cnt[i];
  part[tid] = s;
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int i = 0; i < 1024; ++i) { const int v = part[i]; part[i] = run; run += v; }
  }
  __syncthreads();
  int run = part[tid];
  for (int i = tid * per; i < (tid + 1) * per && i < total; ++i) {
    const int v = cnt[i];
    cnt[i] = run;
    run += v;
  }
  __syncthreads();
  if (tid < W) counts[tid] = cnt[(tid + 1) * nblk] - cnt[tid * nblk];  // (bucket W, the empty ones, follows)
}

__global__ __launch_bounds__(kBlock) void route_place_kernel(
    const int64_t *__restrict__ idx, const int64_t *__restrict__ field_off, int64_t n, int F, int W,
    int64_t per_block, const int *__restrict__ base /* [W][nblk] exclusive */,
    int64_t *__restrict__ pos, int64_t *__restrict__ send_ids, int64_t cap,
    int32_t *__restrict__ overflow) {
  __shared__ int run[kMaxW];          // running offset of each bucket within this block
  __shared__ int wcnt[kBlock / 64][kMaxW];
  // (empty occurrences - idx < 0 - belong to no bucket here: w = -1, pos = -1)
  // cap > 0 (fixed-capacity layout): bucket w starts at w*cap instead of at the packed offset
  if ((int)threadIdx.x < W) {
    const int w = threadIdx.x;
    int r = base[w * gridDim.x + blockIdx.x];
    if (cap > 0) r = r - base[w * gridDim.x] + (int)(w * cap);
    run[w] = r;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t o0 = (int64_t)blockIdx.x * per_block;
  const int64_t o1 = o0 + per_block < n ? o0 + per_block : n;
  for (int64_t ob = o0; ob < o1; ob += kBlock) {
    const int64_t o = ob + threadIdx.x;
    const bool valid = o < o1;
    int64_t g = 0;
    int w = -1;
    if (valid && idx[o] >= 0) {
      g = field_off[o % F] + idx[o];
      w = (int)(g % W);
    }
    int rank_in_wave = 0, my_wave_total = 0;
    for (int q = 0; q < W; ++q) {
      const unsigned long long m = __ballot(w == q);
      if (w == q) rank_in_wave = __popcll(m & ((1ull << lane) - 1ull));
      if (lane == 0) wcnt[wave][q] = __popcll(m);
    }
    (void)my_wave_total;
    __syncthreads();
    if (valid && w < 0) pos[o] = -1;
    if (w >= 0) {
      int off = run[w];
      for (int v = 0; v < wave; ++v) off += wcnt[v][w];
      int64_t p = off + rank_in_wave;
      if (cap > 0 && p >= (int64_t)(w + 1) * cap) {  // bucket over capacity: flag it, stay in bounds
        *overflow = 1;
        p = (int64_t)(w + 1) * cap - 1;
      }
      pos[o] = p;
      send_ids[p] = g / W;
    }
    __syncthreads();
    if ((int)threadIdx.x < W) {
      int t = 0;
      for (int v = 0; v < kBlock / 64; ++v) t += wcnt[v][threadIdx.x];
      run[threadIdx.x] += t;
    }
    __syncthreads();
  }
}

// bucketed[pos[o]][0..D) = d_rows[o][:], [D] = g_bias ? g_bias[b] : 0, [D+1] = g_lin ? g_lin[b] : 0
__global__ __launch_bounds__(kBlock) void pack_grad_rows_kernel(
    const float4 *__restrict__ d_rows, const float *__restrict__ g_bias,
    const float *__restrict__ g_lin, const float *__restrict__ lin_mask, const int64_t *__restrict__ pos,
    int64_t n, int F, int GD, int GW, float4 *__restrict__ out) {
  const int64_t total = n * GW;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t o = t / GW;
    const int sub = (int)(t - o * GW);
    if (pos[o] < 0) continue;  // an empty occurrence: no slot
    float4 v;
    if (sub < GD) {
      v = d_rows[o * GD + sub];
    } else if (sub == GD) {
      const int64_t b = o / F;
      const float lm = lin_mask ? lin_mask[o - b * F] : 1.f;  // linear_features subsets: per field
      v = make_float4(g_bias ? g_bias[b] : 0.f, g_lin ? g_lin[b] * lm : 0.f, 0.f, 0.f);
    } else {
      v = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    out[pos[o] * GW + sub] = v;
  }
}

}  // namespace

extern "C" int64_t rm_shard_route_workspace(int world) { return (int64_t)(world + 1) * 1024 + 64; }

static int shard_route_impl(const int64_t *idx, const int64_t *field_off, int64_t B, int F, int world,
                            int64_t cap, int64_t *pos, int64_t *send_ids, int64_t *counts,
                            int32_t *overflow, int32_t *workspace, rm_stream_t stream) {
  RM_REQUIRE(B >= 0 && F > 0 && world >= 1 && world <= kMaxW, "rm_shard_route: bad sizes (world <= %d)", kMaxW);
  RM_REQUIRE(idx && field_off && pos && send_ids && counts && workspace, "rm_shard_route: NULL argument");
  const int64_t n = B * F;
  RM_REQUIRE(n < (1ll << 31) && cap * world < (1ll << 31), "rm_shard_route: too many occurrences");
  hipStream_t st = (hipStream_t)stream;
  const int nblk = (int)(n == 0 ? 1 : n < 1024 * 256 ? (n + 255) / 256 : 1024);
  const int64_t per_block = ((n + nblk - 1) / nblk + kBlock - 1) / kBlock * kBlock;
  hipLaunchKernelGGL(route_count_kernel, dim3(nblk), dim3(kBlock), 0, st, idx, field_off, n, F, world,
                     per_block, workspace, cap > 0 ? send_ids : nullptr, cap > 0 ? world * cap : 0);
  hipLaunchKernelGGL(route_scan_kernel, dim3(1), dim3(1024), 0, st, workspace, (world + 1) * nblk, nblk, world,
                     counts);
  hipLaunchKernelGGL(route_place_kernel, dim3(nblk), dim3(kBlock), 0, st, idx, field_off, n, F, world,
                     per_block, workspace, pos, send_ids, cap, overflow);
  RM_CHECK_LAUNCH("rm_shard_route");
  return RM_OK;
}

extern "C" int rm_shard_route(const int64_t *idx, const int64_t *field_off, int64_t B, int F,
                              int world, int64_t *pos, int64_t *send_ids, int64_t *counts,
                              int32_t *workspace, rm_stream_t stream) {
  return shard_route_impl(idx, field_off, B, F, world, 0, pos, send_ids, counts, nullptr, workspace, stream);
}

extern "C" int rm_shard_route_padded(const int64_t *idx, const int64_t *field_off, int64_t B, int F,
                                     int world, int64_t cap, int64_t *pos, int64_t *send_ids,
                                     int64_t *counts, int32_t *overflow, int32_t *workspace,
                                     rm_stream_t stream) {
  RM_REQUIRE(cap > 0 && overflow, "rm_shard_route_padded: cap > 0 and an overflow flag are required");
  return shard_route_impl(idx, field_off, B, F, world, cap, pos, send_ids, counts, overflow, workspace,
                          stream);
}

extern "C" int rm_pack_grad_rows(const float *d_rows, const float *g_bias, const float *g_lin,
                                 const float *lin_field_mask, const int64_t *pos, int64_t B, int F, int D,
                                 int width, float *out, rm_stream_t stream) {
  RM_REQUIRE(B >= 0 && F > 0 && D > 0 && D % 4 == 0 && width >= D + 4 && width % 4 == 0,
             "rm_pack_grad_rows: bad sizes (width >= D + 4, multiples of 4)");
  if (B == 0) return RM_OK;
  RM_REQUIRE(d_rows && pos && out && rm_aligned16(d_rows) && rm_aligned16(out),
             "rm_pack_grad_rows: NULL or unaligned argument");
  const int64_t n = B * F;
  const int64_t total = n * (width / 4);
  hipLaunchKernelGGL(pack_grad_rows_kernel, dim3(rm_grid_cap((total + kBlock - 1) / kBlock, 256 * 16)),
                     dim3(kBlock), 0, (hipStream_t)stream, (const float4 *)d_rows, g_bias, g_lin, lin_field_mask,
                     pos, n, F, D / 4, width / 4, (float4 *)out);
  RM_CHECK_LAUNCH("rm_pack_grad_rows");
  return RM_OK;
}
