// clock_probe.hip -- what shader clock does a short, low-occupancy kernel actually run at?
// dependent-FMA chain timed with s_memtime (shader cycles) and s_memrealtime (100 MHz).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(float* out, unsigned long long* t, int iters) {
  float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f;
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 64; ++k) a = fmaf(a, b, c);
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a;
  if (threadIdx.x == 0) { t[2 * blockIdx.x] = c1 - c0; t[2 * blockIdx.x + 1] = r1 - r0; }
}
int main() {
  float* out; unsigned long long* t;
  hipMalloc(&out, 1 << 22); hipMalloc(&t, 1 << 16);
  unsigned long long h[2];
  for (int grid : {1, 256, 1024}) for (int iters : {16, 256, 4096}) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0); hipLaunchKernelGGL(probe, grid, 64, 0, 0, out, t, iters); hipEventRecord(e1);
      hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
    double nfma = 64.0 * iters;
    printf("grid %4d iters %5d: event %.1f us | shader cycles %llu (%.2f cyc/fma) | realtime %.1f us -> clock %.0f MHz\n",
           grid, iters, ms * 1e3, h[0], h[0] / nfma, h[1] / 100.0, h[0] / (h[1] / 100.0));
  }
  return 0;
}
