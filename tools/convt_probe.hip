// Phase timing of the persistent MFMA transposed-conv forward kernel (workgroup 0, wavefront 0).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=fast -DCONVT_PROBE -I vae-gp-ode_amd/csrc tools/convt_probe.hip -o /tmp/convt_probe
#include <cstdio>
#include <vector>
#include <hip/hip_runtime.h>
#include "../vae-gp-ode_amd/csrc/vae_conv_tiled.hip"
namespace gp {
char* error_slot() { static thread_local char e[512]; return e; }
}
template <class L> void run(const char* name, int B) {
  using namespace gp;
  float *x, *w, *b, *y;
  hipMalloc(&x, sizeof(float) * B * L::CI * L::HI * L::HI); hipMalloc(&w, sizeof(float) * L::CI * L::CO * L::K * L::K);
  hipMalloc(&b, sizeof(float) * L::CO); hipMalloc(&y, sizeof(float) * B * L::CO * L::HO * L::HO);
  hipMemset(x, 0, sizeof(float) * B * L::CI * L::HI * L::HI); hipMemset(w, 0, sizeof(float) * L::CI * L::CO * L::K * L::K); hipMemset(b, 0, sizeof(float) * L::CO);
  for (int rep = 0; rep < 3; ++rep) {
    unsigned long long z[8] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(g_probe), z, sizeof(z));
    tiled_bwd_data(x, w, b, y, B, L::CO, L::HO, L::HO, L::CI, L::K, L::S, L::P, L::HI, L::HI, nullptr, 0);
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(z, HIP_SYMBOL(g_probe), sizeof(z));
    if (rep == 2)
      printf("%s B=%d cycles: wstage+range %llu barrier %llu scatter %llu prefetch %llu mma %llu store %llu jobsetup %llu total %llu\n", name, B, z[0], z[1], z[2], z[7], z[3], z[4], z[6], z[5]);
  }
  hipFree(x); hipFree(w); hipFree(b); hipFree(y);
}
template <class L> void runb(const char* name, int B) {
  using namespace gp;
  float *gy, *w, *gx;
  hipMalloc(&gy, sizeof(float) * B * L::CO * L::HO * L::HO); hipMalloc(&w, sizeof(float) * L::CI * L::CO * L::K * L::K);
  hipMalloc(&gx, sizeof(float) * B * L::CI * L::HI * L::HI);
  hipMemset(gy, 0, sizeof(float) * B * L::CO * L::HO * L::HO); hipMemset(w, 0, sizeof(float) * L::CI * L::CO * L::K * L::K);
  for (int rep = 0; rep < 3; ++rep) {
    unsigned long long z[8] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(g_probe), z, sizeof(z));
    tiled_fwd(gy, w, nullptr, gx, B, L::CO, L::HO, L::HO, L::CI, L::K, L::S, L::P, L::HI, L::HI, 0);
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(z, HIP_SYMBOL(g_probe), sizeof(z));
    if (rep == 2)
      printf("%s B=%d cycles: wstage+range %llu barrier %llu scatter %llu prefetch %llu mma %llu store %llu jobsetup %llu total %llu\n", name, B, z[0], z[1], z[2], z[7], z[3], z[4], z[6], z[5]);
  }
  hipFree(gy); hipFree(w); hipFree(gx);
}
template <class L> void runw(const char* name, int B) {
  using namespace gp;
  float *x, *gy, *gw, *scr;
  hipMalloc(&gy, sizeof(float) * B * L::CO * L::HO * L::HO); hipMalloc(&x, sizeof(float) * B * L::CI * L::HI * L::HI);
  hipMalloc(&gw, sizeof(float) * L::CI * L::CO * L::K * L::K); hipMalloc(&scr, sizeof(float) * 300 * L::CI * L::CO * L::K * L::K);
  hipMemset(gy, 0, sizeof(float) * B * L::CO * L::HO * L::HO); hipMemset(x, 0, sizeof(float) * B * L::CI * L::HI * L::HI);
  for (int rep = 0; rep < 3; ++rep) {
    unsigned long long z[8] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(g_probe), z, sizeof(z));
    tiled_bwd_weight(gy, x, gw, scr, B, L::CO, L::HO, L::HO, L::CI, L::K, L::S, L::P, L::HI, L::HI, nullptr, 0);
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(z, HIP_SYMBOL(g_probe), sizeof(z));
    if (rep == 2) printf("%s B=%d cycles: barrier %llu scatter %llu mma %llu total %llu\n", name, B, z[1], z[2], z[3], z[5]);
  }
  hipFree(gy); hipFree(x); hipFree(gw); hipFree(scr);
}
int main() {
  runw<gp::Dec7>("dec7 wgrad", 4096);
  runw<gp::Dec4>("dec4 wgrad", 4096);
  runw<gp::Dec1>("dec1 wgrad", 4096);
  run<gp::Dec7>("dec7", 4096);
  run<gp::Dec4>("dec4", 4096);
  run<gp::Dec1>("dec1", 4096);
  runb<gp::Dec7>("dec7 bwd-data", 4096);
  runb<gp::Dec4>("dec4 bwd-data", 4096);
  runb<gp::Dec1>("dec1 bwd-data", 4096);
  return 0;
}
