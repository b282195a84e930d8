// vae_wgrad_v2.hip -- launchers of the producer / consumer weight-gradient engine (conv_wgrad_v2.hpp) for decnn.7 and decnn.4.
#include <hip/hip_runtime.h>
#include <cstdint>
#include "gp_launch.hpp"
#include "conv_layers.hpp"
#include "conv_wgrad_v2.hpp"

namespace gp {

template <class L, bool PIPE>
static int launch_wgrad_v2(const float* x, const float* gy, float* gw, float* scratch, int B, hipStream_t st, const float* in_bn) {
  constexpr size_t lds = wgrad_v2_lds_bytes<L>();
  static_assert(lds <= 160 * 1024, "LDS budget");
  auto km = k_convT_wgrad_v2<L, false, PIPE>;
  auto kb = k_convT_wgrad_v2<L, true, PIPE>;
  if (set_max_lds((const void*)km, lds) || set_max_lds((const void*)kb, lds)) return 1;
  const int nwg = B < num_cus() ? B : num_cus();
  if (in_bn) hipLaunchKernelGGL(kb, nwg, 768, lds, st, x, gy, scratch, B, in_bn);
  else hipLaunchKernelGGL(km, nwg, 768, lds, st, x, gy, scratch, B, in_bn);
  const size_t n = (size_t)L::CI * L::CO * L::K * L::K;
  if (reduce_job(RedJob{scratch, gw, nwg, (int)n, 1, L::CI / 16, L::CO / 16, L::K * L::K}, st)) return 1;
  return check_launch("convT_wgrad_v2");
}

int wgrad_v2_dec7(const float* x, const float* gy, float* gw, float* scratch, int B, hipStream_t st, const float* in_bn) {
  return launch_wgrad_v2<Dec7, true>(x, gy, gw, scratch, B, st, in_bn);
}
int wgrad_v2_dec4(const float* x, const float* gy, float* gw, float* scratch, int B, hipStream_t st, const float* in_bn) {
  return launch_wgrad_v2<Dec4, true>(x, gy, gw, scratch, B, st, in_bn);
}

}  // namespace gp
