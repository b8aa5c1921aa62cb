// Which f32-input MFMA shape holds the higher clock on MI355X?  Bare loops, operands in registers (random data),
// two waves per SIMD, same output tile per wave (32 rows x 224 columns = 112 accumulator registers):
//   A: 7 x v_mfma_f32_32x32x2_f32 per k2-step (64 cycles each)
//   B: 28 x v_mfma_f32_16x16x4_f32 per k4-step (32 cycles each) - the same FLOPs per 896 cycles
// Prints TFLOP/s and the in-kernel clock (s_memtime / s_memrealtime).  hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512, 1) void loop32(const float *in, float *out, int iters, unsigned long long *clk) {
  const int tid = threadIdx.x;
  float a[4], b[4][7];
  for (int i = 0; i < 4; ++i) {
    a[i] = in[(tid * 4 + i) & 4095];
    for (int j = 0; j < 7; ++j) b[i][j] = in[(tid * 28 + i * 7 + j + 1000) & 4095];
  }
  f32x16 acc[7];
  for (int j = 0; j < 7; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int j = 0; j < 7; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s][j], acc[j], 0, 0, 0);
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float sum = 0.f;
  for (int j = 0; j < 7; ++j)
    for (int r = 0; r < 16; ++r) sum += acc[j][r];
  out[blockIdx.x * 512 + tid] = sum;
  if (tid == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

__global__ __launch_bounds__(512, 1) void loop16(const float *in, float *out, int iters, unsigned long long *clk) {
  const int tid = threadIdx.x;
  float a[2][2], b[2][14];
  for (int i = 0; i < 2; ++i) {
    for (int j = 0; j < 2; ++j) a[i][j] = in[(tid * 4 + i * 2 + j) & 4095];
    for (int j = 0; j < 14; ++j) b[i][j] = in[(tid * 28 + i * 14 + j + 1000) & 4095];
  }
  f32x4 acc[2][14];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 14; ++j)
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 2; ++s)  // two k4-steps = the four k2-steps of loop32
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 14; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][i], b[s][j], acc[i][j], 0, 0, 0);
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float sum = 0.f;
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 14; ++j)
      for (int r = 0; r < 4; ++r) sum += acc[i][j][r];
  out[blockIdx.x * 512 + tid] = sum;
  if (tid == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

// loop32 + the GEMM kernels' operand traffic: per k2-step two ds_read_b128 (the 7 B operands) from a 32 KB LDS image,
// LDSR = 1; + VALU fillers per k2-step, NV > 0
template <int LDSR, int NV>
__global__ __launch_bounds__(512, 1) void loop32x(const float *in, float *out, int iters, unsigned long long *clk) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 8192; i += 512) lds[i] = in[i & 4095];
  __syncthreads();
  float a[4];
  for (int i = 0; i < 4; ++i) a[i] = in[(tid * 4 + i) & 4095];
  float4 b0 = *reinterpret_cast<const float4 *>(lds + lane * 4), b1 = *reinterpret_cast<const float4 *>(lds + 256 + lane * 4);
  f32x16 acc[7];
  for (int j = 0; j < 7; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  float filler = a[0];
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      float4 n0 = b0, n1 = b1;
      if (LDSR) {
        const float *p = lds + ((it * 4 + s + 1) & 15) * 512 + lane * 4;
        n0 = *reinterpret_cast<const float4 *>(p);
        n1 = *reinterpret_cast<const float4 *>(p + 256);
      }
      const float bv[7] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z};
#pragma unroll
      for (int j = 0; j < 7; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], bv[j], acc[j], 0, 0, 0);
#pragma unroll
      for (int v = 0; v < NV; ++v) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(filler) : "v"(a[1]));
      b0 = n0; b1 = n1;
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float sum = filler;
  for (int j = 0; j < 7; ++j)
    for (int r = 0; r < 16; ++r) sum += acc[j][r];
  out[blockIdx.x * 512 + tid] = sum;
  if (tid == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

// loop32x<1, 4> + the GEMM kernels' staging traffic: every 8 k2-steps each thread loads WQ float4 of an L2-resident
// "weight" image (1 MB) and writes them to LDS (W = 1); every 16 k2-steps AQ float4 of a streamed HBM buffer and, at the
// end of every 200 k2-steps ("tile"), 112 dword stores to a streamed output (HB = 1)
template <int W, int HB>
__global__ __launch_bounds__(512, 1) void loop32m(const float *in, float *out, int iters, unsigned long long *clk,
                                                  const float4 *wimg, const float4 *abuf, float *cbuf, size_t abuf_n) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 8192];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 16384; i += 512) lds[i] = in[i & 4095];
  __syncthreads();
  float a[4];
  for (int i = 0; i < 4; ++i) a[i] = in[(tid * 4 + i) & 4095];
  float4 b0 = *reinterpret_cast<const float4 *>(lds + lane * 4), b1 = *reinterpret_cast<const float4 *>(lds + 256 + lane * 4);
  f32x16 acc[7];
  for (int j = 0; j < 7; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  float filler = a[0];
  size_t apos = ((size_t)blockIdx.x * 512 + tid);
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; it += 2) {  // two iterations = 8 k2-steps = one weight chunk
    float4 w0{}, w1{}, w2{}, w3{}, x0{}, x1{};
    if (W) {
      const float4 *wp = wimg + ((size_t)(it >> 1) * 2048 + tid) % 65536;
      w0 = wp[0]; w1 = wp[512]; w2 = wp[1024]; w3 = wp[1536];
    }
    if (HB && (it & 2) == 0) {
      x0 = abuf[apos % abuf_n]; x1 = abuf[(apos + 131072) % abuf_n];
      apos += 262144;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const float *p = lds + ((it * 4 + s + 1) & 15) * 512 + lane * 4;
      const float4 n0 = *reinterpret_cast<const float4 *>(p), n1 = *reinterpret_cast<const float4 *>(p + 256);
      const float bv[7] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z};
#pragma unroll
      for (int j = 0; j < 7; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s & 3], bv[j], acc[j], 0, 0, 0);
#pragma unroll
      for (int v = 0; v < 4; ++v) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(filler) : "v"(a[1]));
      b0 = n0; b1 = n1;
    }
    if (W) {
      float4 *d = reinterpret_cast<float4 *>(lds + 8192) + tid;
      d[0] = w0; d[512] = w1; d[1024] = w2; d[1536] = w3;
    }
    if (HB && (it & 2) == 0) filler += x0.x + x1.y;
    if (HB && (it % 50) == 48) {  // a "tile" ends: 112 dword stores per thread
      float *c = cbuf + ((size_t)blockIdx.x * 512 + tid + (size_t)(it / 50) * 131072 * 112) % (abuf_n * 4 - 131072 * 112);
#pragma unroll
      for (int q = 0; q < 112; ++q) c[(size_t)q * 131072] = acc[q / 16][q % 16];
    }
    __syncthreads();
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float sum = filler;
  for (int j = 0; j < 7; ++j)
    for (int r = 0; r < 16; ++r) sum += acc[j][r];
  out[blockIdx.x * 512 + tid] = sum;
  if (tid == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

__global__ void fill_random(float *p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)(i * 2654435761u) ^ (unsigned)(i >> 7);
    x ^= x << 13; x ^= x >> 17; x ^= x << 5;
    p[i] = ((x & 0xffffff) / 16777216.0f - 0.5f) * 4.0f;
  }
}

int main() {
  const int blocks = 256, iters = 40000;
  float *in, *out;
  unsigned long long *clk;
  hipMalloc(&in, 4096 * 4); hipMalloc(&out, blocks * 512 * 4); hipMalloc(&clk, blocks * 16);
  std::vector<float> h(4096);
  srand(1);
  for (auto &v : h) v = (rand() / (float)RAND_MAX - 0.5f) * 2.0f;
  hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice);
  const size_t abuf_n = (size_t)1 << 26;  // 64 M float4 = 1 GiB streamed buffer (reads); the same region takes the stores
  float4 *wimg, *abuf; float *cbuf;
  hipMalloc(&wimg, 65536 * 16 + 2048 * 16 * 4); hipMalloc(&abuf, abuf_n * 16);
  if (getenv("PROBE_ZERO")) {
    hipMemset(wimg, 0, 65536 * 16 + 2048 * 16 * 4); hipMemset(abuf, 0, abuf_n * 16);
  } else {
    hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, reinterpret_cast<float *>(wimg), (size_t)(65536 + 8192) * 4);
    hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, reinterpret_cast<float *>(abuf), abuf_n * 4);
    hipDeviceSynchronize();
  }
  cbuf = reinterpret_cast<float *>(abuf);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<unsigned long long> hc(blocks * 2);
  for (int rep = 0; rep < 2; ++rep)
    for (int which = 0; which < 9; ++which) {
      hipEventRecord(e0);
      if (which == 0) hipLaunchKernelGGL(loop32, dim3(blocks), dim3(512), 0, 0, in, out, iters, clk);
      else if (which == 1) hipLaunchKernelGGL(loop16, dim3(blocks), dim3(512), 0, 0, in, out, iters, clk);
      else if (which == 2) hipLaunchKernelGGL((loop32x<1, 0>), dim3(blocks), dim3(512), 0, 0, in, out, iters, clk);
      else if (which == 3) hipLaunchKernelGGL((loop32x<0, 4>), dim3(blocks), dim3(512), 0, 0, in, out, iters, clk);
      else if (which == 4) hipLaunchKernelGGL((loop32x<1, 4>), dim3(blocks), dim3(512), 0, 0, in, out, iters, clk);
      else if (which == 5) hipLaunchKernelGGL((loop32x<1, 8>), dim3(blocks), dim3(512), 0, 0, in, out, iters, clk);
      else if (which == 6) hipLaunchKernelGGL((loop32m<0, 0>), dim3(blocks), dim3(512), 0, 0, in, out, iters, clk, wimg, abuf, cbuf, abuf_n);
      else if (which == 7) hipLaunchKernelGGL((loop32m<1, 0>), dim3(blocks), dim3(512), 0, 0, in, out, iters, clk, wimg, abuf, cbuf, abuf_n);
      else hipLaunchKernelGGL((loop32m<1, 1>), dim3(blocks), dim3(512), 0, 0, in, out, iters, clk, wimg, abuf, cbuf, abuf_n);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(hc.data(), clk, blocks * 16, hipMemcpyDeviceToHost);
      double ghz = 0; for (int b = 0; b < blocks; ++b) ghz += (double)hc[2 * b] / hc[2 * b + 1] * 0.1; ghz /= blocks;
      // per wave per iteration: 28 x 32x32x2 (4096 flop x 2... = 2*32*32*2) or 56 x 16x16x4 (2*16*16*4)
      const double flop = (double)blocks * 8 * iters * 28 * 2.0 * 32 * 32 * 2;
      printf("%s  %.3f ms  %.1f TFLOP/s  in-kernel clock %.3f GHz  cycles per step-group %.1f (ideal 1792 per SIMD pair)\n",
             which == 0 ? "32x32x2          " : which == 1 ? "16x16x4          " : which == 2 ? "32x32x2 +lds     " : which == 3 ? "32x32x2 +4valu   " : which == 4 ? "32x32x2 +lds+4v  " : which == 5 ? "32x32x2 +lds+8v  " : which == 6 ? "+lds+4v+barrier  " : which == 7 ? "+L2 weight stage " : "+L2 w +HBM a/c   ", ms, flop / ms / 1e9, ghz, ghz * 1e6 * ms / iters);
    }
  return 0;
}
