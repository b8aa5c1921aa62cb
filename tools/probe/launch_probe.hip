#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k_empty(float* p, int flag) { extern __shared__ float sm[]; if (flag == 12345) p[0] = sm[0]; }
int main() {
  float* d; hipMalloc(&d, 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  int cfgs[][3] = {{256, 512, 135168}, {256, 512, 0}, {256, 256, 65536}, {2048, 256, 0}, {256, 64, 0}, {1, 64, 0}};
  for (auto& c : cfgs) {
    hipFuncSetAttribute((const void*)k_empty, hipFuncAttributeMaxDynamicSharedMemorySize, c[2] ? c[2] : 1);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k_empty, dim3(c[0]), dim3(c[1]), c[2], 0, d, 0);
    hipDeviceSynchronize();
    float tot = 0; const int n = 50;
    for (int i = 0; i < n; ++i) {
      hipEventRecord(a, 0); hipLaunchKernelGGL(k_empty, dim3(c[0]), dim3(c[1]), c[2], 0, d, 0); hipEventRecord(b, 0);
      hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); tot += ms;
    }
    // back-to-back: 50 launches between one event pair
    hipEventRecord(a, 0);
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_empty, dim3(c[0]), dim3(c[1]), c[2], 0, d, 0);
    hipEventRecord(b, 0); hipEventSynchronize(b); float ms2; hipEventElapsedTime(&ms2, a, b);
    printf("grid %4d block %3d lds %6d : single %.1f us, back-to-back %.1f us per launch\n", c[0], c[1], c[2], tot / n * 1e3, ms2 / n * 1e3);
  }
  return 0;
}
