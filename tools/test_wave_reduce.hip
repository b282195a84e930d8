// standalone check of wave_reduce.hpp on exact integer data (run on the GPU box)
#include "../vae-gp-ode_amd/csrc/wave_reduce.hpp"
#include <cstdio>
#include <vector>
template <int NV> __global__ void k(const float* in, float* out) {
  float v[NV], o[NV];
  for (int i = 0; i < NV; ++i) v[i] = in[i * 64 + threadIdx.x];
  gp::wave_sum_all<NV>(v, o);
  for (int i = 0; i < NV; ++i) out[i * 64 + threadIdx.x] = o[i];
}
template <int NV> int run() {
  std::vector<float> h(NV * 64), r(NV * 64);
  for (int i = 0; i < NV; ++i) for (int l = 0; l < 64; ++l) h[i * 64 + l] = (float)((i * 131 + l * 7 + (l * l) % 13) % 97 - 40);
  float *di, *dout;
  hipMalloc(&di, h.size() * 4); hipMalloc(&dout, h.size() * 4);
  hipMemcpy(di, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k<NV>, 1, 64, 0, 0, di, dout);
  hipMemcpy(r.data(), dout, h.size() * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < NV; ++i) {
    float s = 0; for (int l = 0; l < 64; ++l) s += h[i * 64 + l];
    for (int l = 0; l < 64; ++l) if (r[i * 64 + l] != s) { if (bad < 5) printf("NV=%d value %d lane %d: got %g want %g\n", NV, i, l, r[i * 64 + l], s); ++bad; }
  }
  printf("NV=%d: %s\n", NV, bad ? "FAIL" : "ok");
  return bad;
}
int main() { int b = 0; b += run<1>(); b += run<2>(); b += run<3>(); b += run<4>(); b += run<5>(); b += run<6>(); b += run<8>(); b += run<12>(); b += run<16>(); return b != 0; }
