// Issue-rate probe for v_mfma_f32_16x16x4_f32: cycles per MFMA for 1/2/4 independent accumulator chains,
// one or two wavefronts per SIMD, with and without an LDS operand fetch per MFMA.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_probe.hip -o tools/mfma_probe.bin
#include <cstdio>
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC, bool LDS>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters) {
  __shared__ float sm[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) sm[i] = 0.001f * i;
  __syncthreads();
  f32x4 acc[NACC];
  for (int a = 0; a < NACC; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
  float av = threadIdx.x * 0.5f, bv = threadIdx.x * 0.25f;
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16 / NACC; ++u)
#pragma unroll
      for (int a = 0; a < NACC; ++a) {
        float x = av;
        if (LDS) x = sm[(threadIdx.x + 64 * (u * NACC + a) + it) & 4095];
        acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, bv, acc[a], 0, 0, 0);
      }
  }
  const long long t1 = clock64();
  float s = 0.f;
  for (int a = 0; a < NACC; ++a) s += acc[a][0] + acc[a][1] + acc[a][2] + acc[a][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int NACC, bool LDS> void run(int threads, float* out, long long* cyc) {
  const int iters = 2000;
  hipLaunchKernelGGL((k<NACC, LDS>), 256, threads, 0, 0, out, cyc, iters);
  hipLaunchKernelGGL((k<NACC, LDS>), 256, threads, 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("chains %d lds %d waves/SIMD %d: %.1f cycles per MFMA per wave\n", NACC, (int)LDS, threads / 256, (double)c / (iters * 16.0));
}
int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 4 * 256 * 512); hipMalloc(&cyc, 8);
  run<1, false>(256, out, cyc); run<2, false>(256, out, cyc); run<4, false>(256, out, cyc);
  run<4, false>(512, out, cyc); run<4, true>(256, out, cyc); run<4, true>(512, out, cyc);
  return 0;
}
