// CrossNet (DCN v1, vector form): all L layers fused, forward and backward.
// The class is ABSENT from the reference (recman/tf/core/DCN.py:7 has the import
// commented out; used at DCN.py:134-137); arithmetic per arXiv 1708.05123 eq. (3).
//
// HBM-bound by design: the forward reads x0 once and writes 4 B (logit) + 4L B
// (the layer scalars s_l) per example; the backward reads x0 (+ the other
// branch's dx) once and writes dx0 once.  No x_l ever goes to memory.
//
// Mapping: 16 lanes own one example (4 examples per 64-lane wave).  Lane `sub`
// holds the float4 slices q = sub + 16 t (t < T) of the d-vector, i.e. elements
// 4q .. 4q+3; a wave-instruction therefore reads 4 x 256 contiguous bytes.
// Dot products reduce over 16 lanes with 4 xor-shuffles; w/b/w_out sit in LDS,
// zero-padded to 64 T floats so padded lanes contribute exactly 0.
#include "rm_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kMaxL = 8;

__device__ __forceinline__ float dot4(const float4 &a, const float4 &b) {
  return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
}

// loads the float4 slice q of x0 = [xe | xd] for example b (zeros past d), branch-free:
// unconditional loads from clamped addresses + selects (a per-lane if/else here makes hipcc
// serialise the slice loads behind s_waitcnt vmcnt(0))
__device__ __forceinline__ float4 load_x0(const float *__restrict__ xe, const float *__restrict__ xd,
                                          int64_t b, int FD, int Dn, int q) {
  const int e0 = 4 * q;
  const bool in_e = e0 + 3 < FD;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (FD > 0) v = *reinterpret_cast<const float4 *>(xe + b * FD + (in_e ? e0 : 0));
  if (__builtin_amdgcn_ballot_w64(!in_e) == 0) return v;  // wave-uniform: every lane inside xe
  float t[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int kk = e0 + c - FD;
    const bool ok = kk >= 0 && kk < Dn;
    const float x = Dn > 0 ? xd[b * Dn + (ok ? kk : 0)] : 0.f;
    t[c] = ok ? x : 0.f;
  }
  return in_e ? v : make_float4(t[0], t[1], t[2], t[3]);
}

// the original per-lane-branch loader: the backward kernel is register-bound and measured
// faster with it (214 vs 248 us) - the branch-free one above costs it extra spills
__device__ __forceinline__ float4 load_x0_branchy(const float *__restrict__ xe, const float *__restrict__ xd,
                                          int64_t b, int FD, int Dn, int q) {
  const int e0 = 4 * q;
  if (e0 + 3 < FD) return *reinterpret_cast<const float4 *>(xe + b * FD + e0);
  float v[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int e = e0 + c;
    v[c] = (e >= FD && e < FD + Dn) ? xd[b * Dn + (e - FD)] : 0.f;
  }
  return make_float4(v[0], v[1], v[2], v[3]);
}

__device__ __forceinline__ void stage_params(float *sm, const float *__restrict__ w,
                                             const float *__restrict__ b,
                                             const float *__restrict__ w_out, int L, int d, int P) {
  // sm layout: w [L][P] | b [L][P] | w_out [P].  All of a thread's global loads are issued
  // before its first LDS write: a load->store loop serialises one L2 round trip per element
  // (23 per thread here), which dominated the kernel at 1024 short blocks.
  constexpr int PER = 8;
  const int total = (2 * L + 1) * P;
  for (int base = 0; base < total; base += kBlock * PER) {
    float v[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int i = base + q * kBlock + threadIdx.x;
      const int r = i / P, e = i - r * P;
      v[q] = 0.f;
      if (i < total && e < d) v[q] = r < L ? w[r * d + e] : (r < 2 * L ? b[(r - L) * d + e] : w_out[e]);
    }
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int i = base + q * kBlock + threadIdx.x;
      if (i < total) sm[i] = v[q];
    }
  }
}

template <int T>
__global__ __launch_bounds__(kBlock, 3) void cross_fwd_kernel(
    const float *__restrict__ xe, const float *__restrict__ xd, int FD, int Dn,
    const float *__restrict__ w, const float *__restrict__ b, const float *__restrict__ w_out,
    int L, int64_t B, float *__restrict__ logit, float *__restrict__ s_out) {
  extern __shared__ float sm[];
  constexpr int P = 64 * T;
  const int d = FD + Dn;
  stage_params(sm, w, b, w_out, L, d, P);
  __syncthreads();
  const float4 *sw = reinterpret_cast<const float4 *>(sm);
  const float4 *sb = reinterpret_cast<const float4 *>(sm + L * P);
  const float4 *so = reinterpret_cast<const float4 *>(sm + 2 * L * P);

  const int lane = threadIdx.x & 63, sub = lane & 15, ex = lane >> 4;
  const int64_t wave = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (kBlock / 64);
  for (int64_t b0 = wave * 4; b0 < B; b0 += nwaves * 4) {
    const int64_t bi = b0 + ex;
    const bool valid = bi < B;
    const int64_t bb = valid ? bi : B - 1;
    float4 x0[T], x[T];
#pragma unroll
    for (int t = 0; t < T; ++t) x[t] = x0[t] = load_x0(xe, xd, bb, FD, Dn, sub + 16 * t);
    for (int l = 0; l < L; ++l) {
      float part = 0.f;
#pragma unroll
      for (int t = 0; t < T; ++t) part += dot4(x[t], sw[l * (P / 4) + sub + 16 * t]);
      const float s = rm_group_sum<16>(part);
      if (s_out != nullptr && valid && sub == 0) s_out[bb * L + l] = s;
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const float4 bl = sb[l * (P / 4) + sub + 16 * t];
        x[t].x = x0[t].x * s + bl.x + x[t].x;
        x[t].y = x0[t].y * s + bl.y + x[t].y;
        x[t].z = x0[t].z * s + bl.z + x[t].z;
        x[t].w = x0[t].w * s + bl.w + x[t].w;
      }
    }
    float part = 0.f;
#pragma unroll
    for (int t = 0; t < T; ++t) part += dot4(x[t], so[sub + 16 * t]);
    const float out = rm_group_sum<16>(part);
    if (valid && sub == 0) logit[bb] = out;
  }
}

// (A dot-product reformulation of this backward - only x0 as vector state, dx0 = sum_j a_j w_j -
// is algebraically exact (6e-14 vs autograd in fp64) but measured SLOWER here, 274 vs 214 us: its
// dynamic layer loops spill SGPRs.  Kept out; the numbers are in profiles/r01_p7_loader_ablation.md.)
template <int T>
__global__ __launch_bounds__(kBlock, 3) void cross_bwd_kernel(
    const float *__restrict__ xe, const float *__restrict__ xd, int FD, int Dn,
    const float *__restrict__ w, const float *__restrict__ b, const float *__restrict__ w_out,
    int L, int64_t B, const float *__restrict__ g, const float *__restrict__ s_in,
    const float *__restrict__ dx_in_e, const float *__restrict__ dx_in_d,
    float *__restrict__ d_xe, float *__restrict__ d_xd, float *__restrict__ coef) {
  extern __shared__ float sm[];
  constexpr int P = 64 * T;
  const int d = FD + Dn;
  stage_params(sm, w, b, w_out, L, d, P);
  __syncthreads();
  const float4 *sw = reinterpret_cast<const float4 *>(sm);
  const float4 *so = reinterpret_cast<const float4 *>(sm + 2 * L * P);
  const int ncoef = 2 * L + 2;

  const int lane = threadIdx.x & 63, sub = lane & 15, ex = lane >> 4;
  const int64_t wave = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (kBlock / 64);
  for (int64_t b0 = wave * 4; b0 < B; b0 += nwaves * 4) {
    const int64_t bi = b0 + ex;
    const bool valid = bi < B;
    const int64_t bb = valid ? bi : B - 1;
    const float gb = g[bb];
    float4 x0[T], dl[T], acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      x0[t] = load_x0_branchy(xe, xd, bb, FD, Dn, sub + 16 * t);
      const float4 wo = so[sub + 16 * t];
      dl[t] = make_float4(gb * wo.x, gb * wo.y, gb * wo.z, gb * wo.w);  // delta_L = g * w_out
      acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // c_l = 1 + sum_{j<l} s_j  (x_l = c_l x0 + sum_{j<l} b_j).  The layer loop is dynamic and
    // the prefix sums are rebuilt from the (L1-resident) s row each time: unrolling kMaxL
    // layers cost 256 VGPRs = 1 wave per SIMD.
    const float *srow = s_in + bb * L;
    float cL = 1.f;
    for (int j = 0; j < L; ++j) cL += srow[j];
    for (int l = L - 1; l >= 0; --l) {
      const float sl = srow[l];
      float cl = 1.f;
      for (int j = 0; j < l; ++j) cl += srow[j];
      float part = 0.f;
#pragma unroll
      for (int t = 0; t < T; ++t) part += dot4(dl[t], x0[t]);
      const float tl = rm_group_sum<16>(part);  // t_l = delta_{l+1} . x0
      if (valid && sub == 0) {
        coef[bb * ncoef + l] = tl * cl;
        coef[bb * ncoef + L + 1 + l] = tl;
      }
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const float4 wl = sw[l * (P / 4) + sub + 16 * t];
        acc[t].x += sl * dl[t].x; acc[t].y += sl * dl[t].y;
        acc[t].z += sl * dl[t].z; acc[t].w += sl * dl[t].w;
        dl[t].x += tl * wl.x; dl[t].y += tl * wl.y;
        dl[t].z += tl * wl.z; dl[t].w += tl * wl.w;
      }
    }
    if (valid && sub == 0) {
      coef[bb * ncoef + L] = gb * cL;
      coef[bb * ncoef + 2 * L + 1] = gb;
    }
    if (valid) {
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const int q = sub + 16 * t, e0 = 4 * q;
        float4 o = make_float4(acc[t].x + dl[t].x, acc[t].y + dl[t].y, acc[t].z + dl[t].z,
                               acc[t].w + dl[t].w);
        if (e0 + 3 < FD) {
          if (dx_in_e != nullptr) {
            const float4 in = *reinterpret_cast<const float4 *>(dx_in_e + bb * FD + e0);
            o.x += in.x; o.y += in.y; o.z += in.z; o.w += in.w;
          }
          *reinterpret_cast<float4 *>(d_xe + bb * FD + e0) = o;
        } else if (d_xd != nullptr) {
          const float v[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) {
            const int e = e0 + cc;
            if (e >= FD && e < d) {
              float r = v[cc];
              if (dx_in_d != nullptr) r += dx_in_d[bb * Dn + (e - FD)];
              d_xd[bb * Dn + (e - FD)] = r;
            }
          }
        }
      }
    }
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

int pick_T(int d) { return (d + 63) / 64; }

}  // namespace

#define RM_CROSS_DISPATCH(T_, KERNEL, ...)                                                   \
  switch (T_) {                                                                              \
    case 1: hipLaunchKernelGGL((KERNEL<1>), grid, dim3(kBlock), smem, st, __VA_ARGS__); break;   \
    case 2: hipLaunchKernelGGL((KERNEL<2>), grid, dim3(kBlock), smem, st, __VA_ARGS__); break;   \
    case 3: hipLaunchKernelGGL((KERNEL<3>), grid, dim3(kBlock), smem, st, __VA_ARGS__); break;   \
    case 4: hipLaunchKernelGGL((KERNEL<4>), grid, dim3(kBlock), smem, st, __VA_ARGS__); break;   \
    case 5: hipLaunchKernelGGL((KERNEL<5>), grid, dim3(kBlock), smem, st, __VA_ARGS__); break;   \
    case 6: hipLaunchKernelGGL((KERNEL<6>), grid, dim3(kBlock), smem, st, __VA_ARGS__); break;   \
    case 7: hipLaunchKernelGGL((KERNEL<7>), grid, dim3(kBlock), smem, st, __VA_ARGS__); break;   \
    case 8: hipLaunchKernelGGL((KERNEL<8>), grid, dim3(kBlock), smem, st, __VA_ARGS__); break;   \
    default: break;                                                                          \
  }

static int cross_check(const char *fn, const float *xe, const float *xd, int FD, int Dn,
                       const float *w, const float *b, const float *w_out, int L, int64_t B) {
  RM_REQUIRE(B >= 0 && FD >= 0 && Dn >= 0 && FD + Dn > 0, "%s: bad sizes", fn);
  RM_REQUIRE(L >= 1 && L <= kMaxL, "%s: L=%d unsupported (1..%d)", fn, L, kMaxL);
  RM_REQUIRE(FD % 4 == 0, "%s: FD=%d must be a multiple of 4", fn, FD);
  RM_REQUIRE(FD + Dn <= 512, "%s: d=%d unsupported (<= 512)", fn, FD + Dn);
  RM_REQUIRE((FD == 0 || (xe && rm_aligned16(xe))) && (Dn == 0 || xd) && w && b && w_out,
             "%s: NULL or unaligned argument", fn);
  return RM_OK;
}

extern "C" int rm_cross_fwd(const float *xe, const float *xd, int FD, int Dn, const float *w,
                            const float *b, const float *w_out, int L, int64_t B, float *logit,
                            float *s_out, rm_stream_t stream) {
  int rc = cross_check("rm_cross_fwd", xe, xd, FD, Dn, w, b, w_out, L, B);
  if (rc != RM_OK) return rc;
  if (B == 0) return RM_OK;
  RM_REQUIRE(logit, "rm_cross_fwd: logit must not be NULL");
  const int T = pick_T(FD + Dn);
  const size_t smem = (size_t)(2 * L + 1) * 64 * T * sizeof(float);
  dim3 grid(rm_grid_cap((B + 15) / 16, 256 * 3));  // persistent: the blocks grid-stride over examples
  hipStream_t st = (hipStream_t)stream;
  RM_CROSS_DISPATCH(T, cross_fwd_kernel, xe, xd, FD, Dn, w, b, w_out, L, B, logit, s_out)
  RM_CHECK_LAUNCH("rm_cross_fwd");
  return RM_OK;
}

extern "C" int rm_cross_bwd(const float *xe, const float *xd, int FD, int Dn, const float *w,
                            const float *b, const float *w_out, int L, int64_t B, const float *g,
                            const float *s, const float *dx_in_e, const float *dx_in_d,
                            float *d_xe, float *d_xd, float *coef, rm_stream_t stream) {
  int rc = cross_check("rm_cross_bwd", xe, xd, FD, Dn, w, b, w_out, L, B);
  if (rc != RM_OK) return rc;
  if (B == 0) return RM_OK;
  RM_REQUIRE(g && s && coef, "rm_cross_bwd: g, s and coef must not be NULL");
  RM_REQUIRE(FD == 0 || (d_xe && rm_aligned16(d_xe)), "rm_cross_bwd: d_xe NULL or unaligned");
  RM_REQUIRE(!dx_in_e || rm_aligned16(dx_in_e), "rm_cross_bwd: dx_in_e unaligned");
  const int T = pick_T(FD + Dn);
  const size_t smem = (size_t)(2 * L + 1) * 64 * T * sizeof(float);
  dim3 grid(rm_grid_cap((B + 15) / 16, 256 * 3));
  hipStream_t st = (hipStream_t)stream;
  RM_CROSS_DISPATCH(T, cross_bwd_kernel, xe, xd, FD, Dn, w, b, w_out, L, B, g, s, dx_in_e, dx_in_d,
                    d_xe, d_xd, coef)
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
