// capi.hip -- extern "C" surface of libgpode_hip.so (declared in include/gpode.h).
#include "../../include/gpode.h"
#include "gp_launch.hpp"

namespace gp {
char* error_slot() {
  static thread_local char buf[512] = {0};
  return buf;
}
}  // namespace gp

extern "C" {

const char* gpode_version(void) { return "gpode-hip 0.1 (gfx950)"; }
const char* gpode_last_error(void) { return gp::error_slot(); }
int gpode_supported(int kernel, int Di, int Do) { return gp::dims_supported(kernel, Di, Do); }

int gpode_cache_sizes(int kernel, int Di, int Do, int M, int S, size_t* pack_floats, size_t* ws_floats) {
  return gp::cache_sizes(kernel, Di, Do, M, S, pack_floats, ws_floats, 1);
}
int gpode_cache_sizes_n(int kernel, int Di, int Do, int M, int S, int ndraws, size_t* pack_floats, size_t* ws_floats) {
  return gp::cache_sizes(kernel, Di, Do, M, S, pack_floats, ws_floats, ndraws);
}

int gpode_cache_build_fwd_n(int kernel, int Di, int Do, int M, int S, int ndraws,
                            const float* raw_ell, const float* raw_var, const float* Z,
                            const float* Um, const float* Us_packed,
                            const float* eps_u, const float* rff_w, const float* rff_eps, const float* rff_u,
                            float* pack, float* ws,
                            float* ell, float* var, float* omega, float* phase, float* u,
                            float* Lu, float* nu, float* u_prior, void* stream) {
  if (!raw_ell || !raw_var || !Z || !Um || !Us_packed || !eps_u || !rff_w || !rff_eps || !rff_u || !pack || !ws)
    return gp::set_error("gpode_cache_build_fwd: null required pointer");
  return gp::cache_build_fwd(kernel, Di, Do, M, S, ndraws, raw_ell, raw_var, Z, Um, Us_packed, eps_u, rff_w, rff_eps, rff_u,
                             pack, ws, ell, var, omega, phase, u, Lu, nu, u_prior, (hipStream_t)stream);
}
int gpode_cache_build_fwd(int kernel, int Di, int Do, int M, int S,
                          const float* raw_ell, const float* raw_var, const float* Z,
                          const float* Um, const float* Us_packed,
                          const float* eps_u, const float* rff_w, const float* rff_eps, const float* rff_u,
                          float* pack, float* ws,
                          float* ell, float* var, float* omega, float* phase, float* u,
                          float* Lu, float* nu, float* u_prior, void* stream) {
  return gpode_cache_build_fwd_n(kernel, Di, Do, M, S, 1, raw_ell, raw_var, Z, Um, Us_packed, eps_u, rff_w, rff_eps, rff_u, pack, ws, ell, var,
                                 omega, phase, u, Lu, nu, u_prior, stream);
}

int gpode_cache_info(const float* ws, int* host_info, void* stream) {
  // copies the factorisation status word (bit 0: K_uu + jitter I not positive definite) to the host;
  // synchronises `stream`.
  hipError_t e = hipMemcpyAsync(host_info, ws, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream);
  if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
  if (e != hipSuccess) return gp::set_error("gpode_cache_info: %s", hipGetErrorString(e));
  return 0;
}

int gpode_cache_pivots(const float* ws, float* host_min_max, void* stream) {
  // smallest and largest diagonal entry of the Cholesky factor(s) of this draw; synchronises `stream`
  hipError_t e = hipMemcpyAsync(host_min_max, ws + 1, 2 * sizeof(float), hipMemcpyDeviceToHost, (hipStream_t)stream);
  if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
  if (e != hipSuccess) return gp::set_error("gpode_cache_pivots: %s", hipGetErrorString(e));
  return 0;
}

int gpode_set_backward_solves(int mode) {
  if (mode < 0 || mode > 2) return gp::set_error("gpode_set_backward_solves: mode %d (0 auto, 1 always, 2 never)", mode);
  gp::set_backward_solves(mode);
  return 0;
}

int gpode_rhs_fwd(int kernel, int Di, int Do, int M, int S, const float* pack,
                  const float* x, int N, float* f, int mode, void* stream) {
  if (N < 0 || mode < 0 || mode > 2) return gp::set_error("gpode_rhs_fwd: N=%d mode=%d", N, mode);
  if (N == 0) return 0;                              // empty minibatch: nothing to do (empty tensors carry null pointers)
  if (!pack || !x || !f) return gp::set_error("gpode_rhs_fwd: null pointer");
  return gp::rhs_fwd(kernel, Di, Do, M, S, pack, x, N, f, mode, (hipStream_t)stream);
}

static gp::Draws draws_of(int nd, size_t pack, size_t in, size_t out, size_t in2, size_t out2) {
  gp::Draws d;
  d.nd = nd; d.pack = pack; d.in = in; d.out = out; d.in2 = in2; d.out2 = out2;
  return d;
}
static int stage_count(int method) { return method == 0 ? 1 : (method == 1 ? 4 : 2); }

int gpode_rollout_fwd_n(int kernel, int order, int method, int Di, int Do, int M, int S, int ndraws,
                        const float* pack, const float* z0, const float* ts, int N, int T,
                        float* zt, float* xstage, void* stream) {
  if (N < 0 || T < 1 || ndraws < 1 || ndraws > 65535) return gp::set_error("gpode_rollout_fwd: N=%d T=%d draws=%d", N, T, ndraws);
  if (N == 0) return 0;
  if (!pack || !z0 || !ts || !zt) return gp::set_error("gpode_rollout_fwd: null pointer");
  size_t pf = 0;
  if (gp::cache_sizes(kernel, Di, Do, M, S, &pf, nullptr)) return 1;
  return gp::rollout_fwd(kernel, order, method, Di, Do, M, S, pack, z0, ts, N, T, zt, xstage, (hipStream_t)stream,
                         draws_of(ndraws, pf, 0, (size_t)N * T * Di, 0, (size_t)N * (T - 1) * stage_count(method) * Di));
}
int gpode_rollout_fwd(int kernel, int order, int method, int Di, int Do, int M, int S,
                      const float* pack, const float* z0, const float* ts, int N, int T,
                      float* zt, float* xstage, void* stream) {
  return gpode_rollout_fwd_n(kernel, order, method, Di, Do, M, S, 1, pack, z0, ts, N, T, zt, xstage, stream);
}

int gpode_rollout_bwd_n(int kernel, int order, int method, int Di, int Do, int M, int S, int ndraws,
                        const float* pack, const float* xstage, const float* gzt, const float* ts, int N, int T,
                        float* gz0, float* astage, void* stream) {
  if (N < 0 || T < 1 || ndraws < 1 || ndraws > 65535) return gp::set_error("gpode_rollout_bwd: N=%d T=%d draws=%d", N, T, ndraws);
  if (N == 0) return 0;
  if (!pack || !gzt || !ts || !gz0 || (T > 1 && (!xstage || !astage))) return gp::set_error("gpode_rollout_bwd: null pointer");
  size_t pf = 0;
  if (gp::cache_sizes(kernel, Di, Do, M, S, &pf, nullptr)) return 1;
  const size_t rows = (size_t)N * (T - 1) * stage_count(method);
  return gp::rollout_bwd(kernel, order, method, Di, Do, M, S, pack, xstage, gzt, ts, N, T, gz0, astage, (hipStream_t)stream,
                         draws_of(ndraws, pf, rows * Di, (size_t)N * Di, (size_t)N * T * Di, rows * Do));
}
int gpode_rollout_bwd_pgrad_chunks(int kernel, int order, int method, int Di, int Do, int M, int S, int N) {
  return gp::rollout_bwd_pgrad_chunks(kernel, order, method, Di, Do, M, S, N);
}
int gpode_rollout_bwd_pgrad_n(int kernel, int order, int method, int Di, int Do, int M, int S, int ndraws,
                              const float* pack, const float* xstage, const float* gzt, const float* ts, int N, int T,
                              float* gz0, float* astage, float* slab, int nchunk, float* gpack, void* stream) {
  if (N < 1 || T < 2 || ndraws < 1 || ndraws > 65535) return gp::set_error("gpode_rollout_bwd_pgrad: N=%d T=%d draws=%d", N, T, ndraws);
  if (!pack || !gzt || !ts || !gz0 || !xstage || !astage || !slab || !gpack) return gp::set_error("gpode_rollout_bwd_pgrad: null pointer");
  size_t pf = 0;
  if (gp::cache_sizes(kernel, Di, Do, M, S, &pf, nullptr)) return 1;
  const size_t rows = (size_t)N * (T - 1) * stage_count(method);
  return gp::rollout_bwd_pgrad(kernel, order, method, Di, Do, M, S, pack, xstage, gzt, ts, N, T, gz0, astage, slab, nchunk, gpack,
                               (hipStream_t)stream, draws_of(ndraws, pf, rows * Di, (size_t)N * Di, (size_t)N * T * Di, rows * Do));
}
int gpode_rollout_bwd(int kernel, int order, int method, int Di, int Do, int M, int S,
                      const float* pack, const float* xstage, const float* gzt, const float* ts, int N, int T,
                      float* gz0, float* astage, void* stream) {
  return gpode_rollout_bwd_n(kernel, order, method, Di, Do, M, S, 1, pack, xstage, gzt, ts, N, T, gz0, astage, stream);
}

int gpode_rhs_vjp(int kernel, int Di, int Do, int M, int S, const float* pack,
                  const float* x, const float* a, int R, float* gx, void* stream) {
  if (!pack || !x || !a || !gx) return gp::set_error("gpode_rhs_vjp: null pointer");
  return gp::rhs_vjp(kernel, Di, Do, M, S, pack, x, a, R, gx, 0, (hipStream_t)stream);
}

int gpode_param_grad_n(int kernel, int Di, int Do, int M, int S, int ndraws, const float* pack,
                       const float* x, const float* a, int R, float* slab, int nchunk, float* gpack, int accumulate,
                       void* stream) {
  if (!pack || !x || !a || !slab || !gpack) return gp::set_error("gpode_param_grad: null pointer");
  if (ndraws < 1 || ndraws > 65535) return gp::set_error("gpode_param_grad: %d draws", ndraws);
  size_t pf = 0;
  if (gp::cache_sizes(kernel, Di, Do, M, S, &pf, nullptr)) return 1;
  return gp::param_grad(kernel, Di, Do, M, S, pack, x, a, R, slab, nchunk, gpack, accumulate, 0, (hipStream_t)stream,
                        draws_of(ndraws, pf, (size_t)R * Di, pf, (size_t)R * Do, 0));
}
int gpode_param_grad(int kernel, int Di, int Do, int M, int S, const float* pack,
                     const float* x, const float* a, int R, float* slab, int nchunk, float* gpack, int accumulate,
                     void* stream) {
  return gpode_param_grad_n(kernel, Di, Do, M, S, 1, pack, x, a, R, slab, nchunk, gpack, accumulate, stream);
}

int gpode_kernel_matrix(int kernel, int Di, int Do, const float* raw_ell, const float* raw_var,
                        const float* X, int N, const float* X2, int M2, float* out, void* stream) {
  if (!raw_ell || !raw_var || !X || !X2 || !out) return gp::set_error("gpode_kernel_matrix: null pointer");
  return gp::kernel_matrix(kernel, Di, Do, raw_ell, raw_var, X, N, X2, M2, out, (hipStream_t)stream);
}

size_t gpode_kern_scratch(int kernel, int Di, int Do, int M, int S) {
  if (!gp::dims_supported(kernel, Di, Do) || M < 0 || S < 0) return 0;
  return gp::kern_scratch_floats(kernel, Di, Do, M, S);
}
int gpode_kern_cache(int kernel, int Di, int Do, int S, const float* raw_ell, const float* raw_var, const float* rff_w,
                     const float* rff_eps, const float* rff_u, float* pack, float* omega, float* phase, void* stream) {
  if (!raw_ell || !raw_var || !rff_w || !rff_eps || !rff_u || !pack) return gp::set_error("gpode_kern_cache: null pointer");
  return gp::kern_cache(kernel, Di, Do, S, raw_ell, raw_var, rff_w, rff_eps, rff_u, pack, omega, phase, (hipStream_t)stream);
}
int gpode_compute_nu_ws(int kernel, int Di, int Do, int M, size_t* ws_floats) {
  if (!ws_floats) return gp::set_error("gpode_compute_nu_ws: null pointer");
  return gp::compute_nu_ws(kernel, Di, Do, M, ws_floats);
}
int gpode_compute_nu(int kernel, int Di, int Do, int M, const float* Ku, const float* u_prior, const float* u, float* nu, float* ws,
                     void* stream) {
  if (!Ku || !u_prior || !u || !nu || !ws) return gp::set_error("gpode_compute_nu: null pointer");
  return gp::compute_nu(kernel, Di, Do, M, Ku, u_prior, u, nu, ws, (hipStream_t)stream);
}
int gpode_f_update(int kernel, int Di, int Do, int M, const float* raw_ell, const float* raw_var, const float* x2, const float* nu,
                   const float* x, int N, float* out, float* scratch, void* stream) {
  if (N < 0) return gp::set_error("gpode_f_update: N=%d", N);
  if (N == 0) return 0;
  if (!raw_ell || !raw_var || !x2 || !nu || !x || !out || !scratch) return gp::set_error("gpode_f_update: null pointer");
  return gp::f_update(kernel, Di, Do, M, raw_ell, raw_var, x2, nu, x, N, out, scratch, (hipStream_t)stream);
}

int gpode_conditional_ws(int Di, int Do, int M, int N, size_t* ws_floats) {
  if (!ws_floats) return gp::set_error("gpode_conditional_ws: null pointer");
  return gp::conditional_ws(Di, Do, M, N, ws_floats);
}

int gpode_conditional(int Di, int Do, int M, int N, const float* raw_ell, const float* raw_var, const float* Z, const float* Um,
                      const float* Us, int us_rank1, const float* x, int full_cov, float* mean, float* var, float* ws, void* stream) {
  if (!raw_ell || !raw_var || !Z || !Um || !Us || !x || !mean || !var || !ws) return gp::set_error("gpode_conditional: null pointer");
  return gp::conditional(Di, Do, M, N, raw_ell, raw_var, Z, Um, Us, us_rank1, x, full_cov, mean, var, ws, (hipStream_t)stream);
}

int gpode_svgp_kl_fwd(int M, int Do, const float* Um, const float* Us_packed, float* kl, void* stream) {
  if (!Um || !Us_packed || !kl) return gp::set_error("gpode_svgp_kl_fwd: null pointer");
  return gp::svgp_kl_fwd(M, Do, Um, Us_packed, kl, (hipStream_t)stream);
}

int gpode_svgp_kl_bwd(int M, int Do, const float* Um, const float* Us_packed, const float* g,
                      float* dUm, float* dUs, void* stream) {
  if (!Um || !Us_packed || !g || !dUm || !dUs) return gp::set_error("gpode_svgp_kl_bwd: null pointer");
  return gp::svgp_kl_bwd(M, Do, Um, Us_packed, g, dUm, dUs, (hipStream_t)stream);
}

int gpode_cache_bwd_sizes_n(int kernel, int Di, int Do, int M, int S, int ndraws, size_t* bws_floats) {
  if (!bws_floats) return gp::set_error("gpode_cache_bwd_sizes: null pointer");
  if (ndraws < 1) return gp::set_error("gpode_cache_bwd_sizes: %d draws", ndraws);
  return gp::cache_bwd_sizes(kernel, Di, Do, M, S, ndraws, bws_floats);
}
int gpode_cache_bwd_sizes(int kernel, int Di, int Do, int M, int S, size_t* bws_floats) {
  return gpode_cache_bwd_sizes_n(kernel, Di, Do, M, S, 1, bws_floats);
}

int gpode_cache_build_bwd_n(int kernel, int Di, int Do, int M, int S, int ndraws,
                            const float* raw_ell, const float* raw_var, const float* Z, const float* eps_u,
                            const float* pack, const float* ws, float* gpack, float* bws,
                            float* g_raw_ell, float* g_raw_var, float* g_Z, float* g_Um, float* g_Us, int prepared, void* stream) {
  if (!raw_ell || !raw_var || !Z || !eps_u || !pack || !ws || !gpack || !bws || !g_raw_ell || !g_raw_var || !g_Z || !g_Um || !g_Us)
    return gp::set_error("gpode_cache_build_bwd: null pointer");
  if (ndraws < 1 || ndraws > 65535) return gp::set_error("gpode_cache_build_bwd: %d draws", ndraws);
  return gp::cache_build_bwd(kernel, Di, Do, M, S, ndraws, raw_ell, raw_var, Z, eps_u, pack, ws, gpack, bws,
                             g_raw_ell, g_raw_var, g_Z, g_Um, g_Us, prepared, (hipStream_t)stream);
}
int gpode_cache_build_bwd(int kernel, int Di, int Do, int M, int S,
                          const float* raw_ell, const float* raw_var, const float* Z, const float* eps_u,
                          const float* pack, const float* ws, float* gpack, float* bws,
                          float* g_raw_ell, float* g_raw_var, float* g_Z, float* g_Um, float* g_Us, int prepared, void* stream) {
  return gpode_cache_build_bwd_n(kernel, Di, Do, M, S, 1, raw_ell, raw_var, Z, eps_u, pack, ws, gpack, bws, g_raw_ell, g_raw_var, g_Z, g_Um,
                                 g_Us, prepared, stream);
}
int gpode_cache_bwd_prepare_n(int kernel, int Di, int Do, int M, int S, int ndraws, const float* ws, float* bws, void* stream) {
  if (!ws || !bws) return gp::set_error("gpode_cache_bwd_prepare: null pointer");
  if (ndraws < 1) return gp::set_error("gpode_cache_bwd_prepare: %d draws", ndraws);
  return gp::cache_bwd_prepare(kernel, Di, Do, M, S, ndraws, ws, bws, (hipStream_t)stream);
}
int gpode_cache_bwd_prepare(int kernel, int Di, int Do, int M, int S, const float* ws, float* bws, void* stream) {
  return gpode_cache_bwd_prepare_n(kernel, Di, Do, M, S, 1, ws, bws, stream);
}

// ---- conv VAE blocks -----------------------------------------------------------------------
#define GP_ST ((hipStream_t)stream)
int gpode_conv2d_fwd(const float* x, const float* w, const float* bias, float* y, int B, int Ci, int H, int W, int Co,
                     int K, int S, int P, int Ho, int Wo, void* stream) {
  if (!x || !w || !y) return gp::set_error("gpode_conv2d_fwd: null pointer");
  return gp::conv2d_fwd(x, w, bias, y, B, Ci, H, W, Co, K, S, P, Ho, Wo, GP_ST);
}
int gpode_conv2d_bwd_data(const float* gy, const float* w, const float* bias, float* gx, int B, int Ci, int H, int W, int Co,
                          int K, int S, int P, int Ho, int Wo, void* stream) {
  if (!gy || !w || !gx) return gp::set_error("gpode_conv2d_bwd_data: null pointer");
  return gp::conv2d_bwd_data(gy, w, bias, gx, B, Ci, H, W, Co, K, S, P, Ho, Wo, nullptr, GP_ST);
}
int gpode_conv2d_bwd_data_bn(const float* gy, const float* gy_bn, const float* w, const float* bias, float* gx, int B, int Ci, int H,
                             int W, int Co, int K, int S, int P, int Ho, int Wo, void* stream) {
  if (!gy || !gy_bn || !w || !gx) return gp::set_error("gpode_conv2d_bwd_data_bn: null pointer");
  return gp::conv2d_bwd_data(gy, w, bias, gx, B, Ci, H, W, Co, K, S, P, Ho, Wo, gy_bn, GP_ST);
}
int gpode_conv2d_fwd_bs(const float* x, size_t x_batch_stride, const float* w, const float* bias, float* y, int B, int Ci, int H, int W, int Co,
                        int K, int S, int P, int Ho, int Wo, void* stream) {
  if (!x || !w || !y) return gp::set_error("gpode_conv2d_fwd_bs: null pointer");
  if (x_batch_stride < (size_t)Ci * H * W) return gp::set_error("gpode_conv2d_fwd_bs: images overlap");
  return gp::conv2d_fwd(x, w, bias, y, B, Ci, H, W, Co, K, S, P, Ho, Wo, GP_ST, x_batch_stride);
}
int gpode_conv2d_bwd_weight_bs(const float* x, size_t x_batch_stride, const float* gy, float* gw, float* gbias, float* scratch, int B, int Ci,
                               int H, int W, int Co, int K, int S, int P, int Ho, int Wo, void* stream) {
  if (!x || !gy || !gw || !scratch) return gp::set_error("gpode_conv2d_bwd_weight_bs: null pointer");
  if (x_batch_stride < (size_t)Ci * H * W) return gp::set_error("gpode_conv2d_bwd_weight_bs: images overlap");
  return gp::conv2d_bwd_weight(x, gy, gw, gbias, scratch, B, Ci, H, W, Co, K, S, P, Ho, Wo, nullptr, GP_ST, x_batch_stride);
}
size_t gpode_convT_fwd_stats_scratch(int Cout) { return gp::convT_fwd_stats_scratch(Cout); }
int gpode_convT_fwd_stats(const float* x, const float* x_bn, const float* w, const float* bias, float* y, int B, int Ci, int H, int W, int Co,
                          int K, int S, int P, int Ho, int Wo, const float* gamma, const float* beta, float* save_mean, float* save_invstd,
                          float* running_mean, float* running_var, long long* num_batches_tracked, float momentum, float eps, float* table,
                          float* scratch, int slot, void* stream) {
  if (!x || !w || !y || !gamma || !beta || !save_mean || !save_invstd || !table || !scratch)
    return gp::set_error("gpode_convT_fwd_stats: null pointer");
  if ((running_mean == nullptr) != (running_var == nullptr)) return gp::set_error("gpode_convT_fwd_stats: running_mean and running_var go together");
  return gp::convT_fwd_stats(x, x_bn, w, bias, y, B, Ci, H, W, Co, K, S, P, Ho, Wo, gamma, beta, save_mean, save_invstd, running_mean,
                             running_var, num_batches_tracked, momentum, eps, table, scratch, slot, GP_ST);
}
size_t gpode_conv_wgrad_scratch(int B, int Ci, int Co, int K) { return gp::conv_wgrad_scratch(B, Ci, Co, K); }
int gpode_conv2d_bwd_weight(const float* x, const float* gy, float* gw, float* gbias, float* scratch, int B, int Ci, int H, int W,
                            int Co, int K, int S, int P, int Ho, int Wo, void* stream) {
  if (!x || !gy || !gw || !scratch) return gp::set_error("gpode_conv2d_bwd_weight: null pointer");
  return gp::conv2d_bwd_weight(x, gy, gw, gbias, scratch, B, Ci, H, W, Co, K, S, P, Ho, Wo, nullptr, GP_ST);
}
int gpode_conv2d_bwd_weight_bn(const float* x, const float* gy, const float* gy_bn, float* gw, float* gbias, float* scratch, int B, int Ci,
                               int H, int W, int Co, int K, int S, int P, int Ho, int Wo, void* stream) {
  if (!x || !gy || !gy_bn || !gw || !scratch) return gp::set_error("gpode_conv2d_bwd_weight_bn: null pointer");
  return gp::conv2d_bwd_weight(x, gy, gw, gbias, scratch, B, Ci, H, W, Co, K, S, P, Ho, Wo, gy_bn, GP_ST);
}
size_t gpode_bn_scratch(int B, int C) { return gp::bn_scratch(B, C); }
int gpode_bn_fwd(const float* x, const float* gamma, const float* beta, float* y, float* save_mean, float* save_invstd,
                 float* running_mean, float* running_var, long long* num_batches_tracked, float momentum, float eps, int B, int C,
                 int HW, int relu, float* scratch, void* stream) {
  if (!x || !gamma || !beta || !y || !save_mean || !save_invstd || !scratch) return gp::set_error("gpode_bn_fwd: null pointer");
  return gp::bn_fwd(x, gamma, beta, y, save_mean, save_invstd, running_mean, running_var, num_batches_tracked, momentum, eps, B, C, HW, relu,
                    scratch, GP_ST);
}
int gpode_bn_bwd(const float* x, const float* gy, const float* gamma, const float* beta, const float* save_mean,
                 const float* save_invstd, float* gx, float* ggamma, float* gbeta, float* gx_chansum, int B, int C, int HW,
                 int relu, float* scratch, void* stream) {
  if (!x || !gy || !gamma || !beta || !save_mean || !save_invstd || !gx || !ggamma || !gbeta || !scratch)
    return gp::set_error("gpode_bn_bwd: null pointer");
  return gp::bn_bwd(x, gy, gamma, beta, save_mean, save_invstd, gx, ggamma, gbeta, gx_chansum, B, C, HW, relu, scratch, GP_ST);
}
int gpode_bn_stats(const float* x, const float* gamma, const float* beta, float* save_mean, float* save_invstd, float* running_mean,
                   float* running_var, long long* num_batches_tracked, float momentum, float eps, float* table, int B, int C, int HW,
                   float* scratch, void* stream) {
  if (!x || !gamma || !beta || !save_mean || !save_invstd || !table || !scratch) return gp::set_error("gpode_bn_stats: null pointer");
  return gp::bn_stats(x, gamma, beta, save_mean, save_invstd, running_mean, running_var, num_batches_tracked, momentum, eps, table, B, C, HW,
                      scratch, GP_ST);
}
int gpode_bn_moments(const float* x, float* moments, int B, int C, int HW, float* scratch, void* stream) {
  if (!x || !moments || !scratch) return gp::set_error("gpode_bn_moments: null pointer");
  return gp::bn_moments(x, moments, B, C, HW, scratch, GP_ST);
}
int gpode_bn_finalize(const float* gathered, int nranks, const float* gamma, const float* beta, float* save_mean, float* save_invstd,
                      float* running_mean, float* running_var, long long* num_batches_tracked, float momentum, float eps, float* table,
                      int C, void* stream) {
  if (!gathered || !gamma || !beta || !save_mean || !save_invstd || !table) return gp::set_error("gpode_bn_finalize: null pointer");
  return gp::bn_finalize(gathered, nranks, gamma, beta, save_mean, save_invstd, running_mean, running_var, num_batches_tracked, momentum, eps,
                         table, C, GP_ST);
}
int gpode_bn_apply(const float* x, const float* table, float* y, int B, int C, int HW, int relu, void* stream) {
  if (!x || !table || !y) return gp::set_error("gpode_bn_apply: null pointer");
  return gp::bn_apply(x, table, y, B, C, HW, relu, GP_ST);
}
int gpode_bn_bwd_sums(const float* x, const float* gy, const float* gamma, const float* beta, const float* save_mean,
                      const float* save_invstd, float* sums, int B, int C, int HW, int relu, float* scratch, void* stream) {
  if (!x || !gy || !gamma || !beta || !save_mean || !save_invstd || !sums || !scratch) return gp::set_error("gpode_bn_bwd_sums: null pointer");
  return gp::bn_bwd_sums(x, gy, gamma, beta, save_mean, save_invstd, sums, B, C, HW, relu, scratch, GP_ST);
}
int gpode_bn_bwd_apply(const float* x, const float* gy, const float* gamma, const float* beta, const float* save_mean,
                       const float* save_invstd, const float* sums_gathered, const float* weights, int nranks, float count_all,
                       float* gx, float* ggamma, float* gbeta, float* gx_chansum, int B, int C, int HW, int relu, float* scratch,
                       void* stream) {
  if (!x || !gy || !gamma || !beta || !save_mean || !save_invstd || !sums_gathered || !weights || !gx || !ggamma || !gbeta || !scratch)
    return gp::set_error("gpode_bn_bwd_apply: null pointer");
  return gp::bn_bwd_apply(x, gy, gamma, beta, save_mean, save_invstd, sums_gathered, weights, nranks, count_all, gx, ggamma, gbeta, gx_chansum, B, C, HW, relu,
                          scratch, GP_ST);
}
void gpode_defer_reductions(int mode) { gp::defer_reductions(mode); }
int gpode_flush_reductions(void* stream) { return gp::flush_reductions(GP_ST); }
int gpode_dec10_bn_scratch_floats(void) { return gp::dec10_bn_scratch_floats(); }
int gpode_dec10_bn_bwd_sums(const float* c, const float* gy, const float* w, const float* gamma, const float* beta, const float* save_mean,
                            const float* save_invstd, float* sums, int B, float* scratch, void* stream) {
  if (!c || !gy || !w || !gamma || !beta || !save_mean || !save_invstd || !scratch) return gp::set_error("gpode_dec10_bn_bwd_sums: null pointer");
  if (B < 1) return gp::set_error("gpode_dec10_bn_bwd_sums: B >= 1");
  return gp::dec10_bn_bwd_sums(c, gy, w, gamma, beta, save_mean, save_invstd, sums, B, scratch, GP_ST);
}
int gpode_dec10_bn_wgrad_scratch_floats(void) { return gp::dec10_bn_wgrad_scratch_floats(); }
int gpode_dec10_bn_bwd_sums_wgrad(const float* c, const float* gy, const float* w, const float* gamma, const float* beta, const float* save_mean,
                                  const float* save_invstd, float* sums, float* gw, float* gbias, int B, float* scratch, float* wscratch,
                                  void* stream) {
  if (!c || !gy || !w || !gamma || !beta || !save_mean || !save_invstd || !scratch || !gw || !wscratch)
    return gp::set_error("gpode_dec10_bn_bwd_sums_wgrad: null pointer");
  if (B < 1) return gp::set_error("gpode_dec10_bn_bwd_sums_wgrad: B >= 1");
  return gp::dec10_bn_bwd_sums(c, gy, w, gamma, beta, save_mean, save_invstd, sums, B, scratch, GP_ST, gw, wscratch, gbias);
}
int gpode_dec10_bn_bwd_apply(const float* c, const float* gy, const float* w, const float* gamma, const float* beta, const float* save_mean,
                             const float* save_invstd, const float* sums_gathered, const float* weights, int nranks, float count_all,
                             float* gc, float* ggamma, float* gbeta, float* gc_chansum, int B, float* scratch, void* stream) {
  if (!c || !gy || !w || !gamma || !beta || !save_mean || !save_invstd || !gc || !ggamma || !gbeta || !scratch)
    return gp::set_error("gpode_dec10_bn_bwd_apply: null pointer");
  if (B < 1) return gp::set_error("gpode_dec10_bn_bwd_apply: B >= 1");
  return gp::dec10_bn_bwd_apply(c, gy, w, gamma, beta, save_mean, save_invstd, sums_gathered, weights, nranks, count_all, gc, ggamma, gbeta, gc_chansum,
                                B, scratch, GP_ST);
}
int gpode_bn_eval(const float* x, const float* gy, const float* gamma, const float* beta, const float* running_mean,
                  const float* running_var, float eps, float* out, int B, int C, int HW, int relu, void* stream) {
  if (!x || !gamma || !beta || !running_mean || !running_var || !out) return gp::set_error("gpode_bn_eval: null pointer");
  return gp::bn_eval(x, gy, gamma, beta, running_mean, running_var, eps, out, B, C, HW, relu, GP_ST);
}
int gpode_chan_sum(const float* v, float* out, int B, int C, int HW, float* scratch, void* stream) {
  if (!v || !out || !scratch) return gp::set_error("gpode_chan_sum: null pointer");
  return gp::chan_sum(v, out, B, C, HW, scratch, GP_ST);
}
int gpode_act_fwd(const float* x, float* y, size_t n, int mode, void* stream) { return gp::act_fwd(x, y, n, mode, GP_ST); }
int gpode_act_bwd(const float* y, const float* gy, float* gx, size_t n, int mode, void* stream) { return gp::act_bwd(y, gy, gx, n, mode, GP_ST); }
int gpode_linear_fwd(const float* x, const float* w, const float* bias, float* y, int B, int In, int Out, void* stream) {
  if (!x || !w || !y) return gp::set_error("gpode_linear_fwd: null pointer");
  return gp::linear_fwd(x, w, bias, y, B, In, Out, GP_ST);
}
size_t gpode_linear_bwd_scratch(int B, int In, int Out) { return gp::linear_bwd_scratch(B, In, Out); }
int gpode_linear_bwd(const float* x, const float* w, const float* gy, float* gx, float* gw, float* gb, int B, int In, int Out,
                     float* scratch, void* stream) {
  if (!x || !w || !gy) return gp::set_error("gpode_linear_bwd: null pointer");
  return gp::linear_bwd(x, w, gy, gx, gw, gb, B, In, Out, scratch, GP_ST);
}
int gpode_linear_relu_fwd(const float* x, const float* w, const float* bias, float* y, int B, int In, int Out, void* stream) {
  if (!x || !w || !y) return gp::set_error("gpode_linear_relu_fwd: null pointer");
  return gp::linear_relu_fwd(x, w, bias, y, B, In, Out, GP_ST);
}
int gpode_linear_relu_bwd(const float* x, const float* w, const float* gy, float* gx, float* gw, float* gb, int B, int In, int Out,
                          void* stream) {
  if (!x || !w || !gy) return gp::set_error("gpode_linear_relu_bwd: null pointer");
  return gp::linear_relu_bwd(x, w, gy, gx, gw, gb, B, In, Out, GP_ST);
}
int gpode_loglik_fwd(const float* X, const float* z, float* ll, size_t n, size_t nX, void* stream) { return gp::loglik_fwd(X, z, ll, n, nX, GP_ST); }
int gpode_loglik_bwd(const float* X, const float* z, const float* g, float* gz, size_t n, size_t nX, void* stream) {
  return gp::loglik_bwd(X, z, g, gz, n, nX, GP_ST);
}
int gpode_loglik_rowsum_fwd(const float* X, const float* z, float* out, size_t rows, size_t inner, size_t nX, void* stream) {
  return gp::loglik_rowsum_fwd(X, z, out, rows, inner, nX, GP_ST);
}
int gpode_loglik_rowsum_bwd(const float* X, const float* z, const float* grow, float* gz, size_t rows, size_t inner, size_t nX,
                            void* stream) {
  return gp::loglik_rowsum_bwd(X, z, grow, gz, rows, inner, nX, GP_ST);
}
int gpode_noise_fill(float* out, long long n_normal, long long n_uniform, unsigned long long seed, unsigned long long* state, void* stream) {
  if (n_normal < 0 || n_uniform < 0) return gp::set_error("gpode_noise_fill: negative count");
  if (n_normal + n_uniform == 0) return 0;
  if (!out || !state) return gp::set_error("gpode_noise_fill: null pointer");
  return gp::noise_fill(out, n_normal, n_uniform, seed, state, GP_ST);
}
int gpode_reparam_kl_fwd(const float* mu, const float* logvar, int ld, const float* eps, float* z, float* klpart, int N, int q, void* stream) {
  if (N == 0) return 0;
  if (!mu || !logvar || !eps || !z || !klpart || q < 1 || ld < q) return gp::set_error("gpode_reparam_kl_fwd: bad argument");
  return gp::reparam_kl_fwd(mu, logvar, ld, eps, z, klpart, N, q, GP_ST);
}
int gpode_reparam_kl_bwd(const float* gz, const float* gklpart, const float* mu, const float* logvar, int ld, const float* eps, float* gmu,
                         float* glogvar, int ldg, int N, int q, void* stream) {
  if (N == 0) return 0;
  if (!mu || !logvar || !eps || !gmu || !glogvar || q < 1 || ld < q || ldg < q) return gp::set_error("gpode_reparam_kl_bwd: bad argument");
  return gp::reparam_kl_bwd(gz, gklpart, mu, logvar, ld, eps, gmu, glogvar, ldg, N, q, GP_ST);
}
int gpode_elbo_all_fwd_kl(const float* lpart, int nl_rows, int nl_values, const float* kls, int nks, const float* klv, int nkv, int N, int M,
                          int Do, const float* Um, const float* Us, float nobs, float* out, void* stream) {
  if (!lpart || !kls || !Um || !Us || !out || (nkv > 0 && !klv)) return gp::set_error("gpode_elbo_all_fwd_kl: null pointer");
  if (nl_rows < 1 || nl_values < nl_rows || N < 1 || M < 1 || Do < 1 || nks < 1 || nkv < 0) return gp::set_error("gpode_elbo_all_fwd_kl: bad sizes");
  return gp::elbo_all_fwd_kl(lpart, nl_rows, nl_values, kls, nks, klv, nkv, N, M, Do, Um, Us, nobs, out, GP_ST);
}
int gpode_elbo_all_bwd_ll_kl(const float* g_loss, const float* g_nll, const float* g_kl, const float* g_klu, int nl_rows, int N, int M, int Do,
                             const float* Um, const float* Us, float nobs, float* glrow, float* gkls, int nks, float* gklv, int nkv,
                             float* dUm, float* dUs, const float* X, const float* z, float* ga, size_t n_logits, size_t nX, void* stream) {
  if (!Um || !Us || !glrow || !gkls || !dUm || !dUs || (nkv > 0 && !gklv) || !X || !z || !ga) return gp::set_error("gpode_elbo_all_bwd_ll_kl: null pointer");
  if (nl_rows < 1 || N < 1 || M < 1 || Do < 1 || nks < 1 || nkv < 0 || n_logits < 1 || nX < 1 || n_logits % nX != 0)
    return gp::set_error("gpode_elbo_all_bwd_ll_kl: bad sizes");
  return gp::elbo_all_bwd_ll_kl(g_loss, g_nll, g_kl, g_klu, nl_rows, N, M, Do, Um, Us, nobs, glrow, gkls, nks, gklv, nkv, dUm, dUs, X, z, ga,
                                n_logits, nX, GP_ST);
}
int gpode_reparam_fwd(const float* mu, const float* logvar, int ld, const float* eps, float* z, int N, int q, void* stream) {
  if (N == 0) return 0;
  if (!mu || !logvar || !eps || !z || q < 1 || ld < q) return gp::set_error("gpode_reparam_fwd: bad argument");
  return gp::reparam_fwd(mu, logvar, ld, eps, z, N, q, GP_ST);
}
int gpode_reparam_bwd(const float* gz, const float* logvar, int ld, const float* eps, float* gmu, float* glogvar, int ldg, int N, int q,
                      void* stream) {
  if (N == 0) return 0;
  if (!gz || !logvar || !eps || !gmu || !glogvar || q < 1 || ld < q || ldg < q) return gp::set_error("gpode_reparam_bwd: bad argument");
  return gp::reparam_bwd(gz, logvar, ld, eps, gmu, glogvar, ldg, N, q, GP_ST);
}
int gpode_normal_kl_fwd(const float* mu, const float* logvar, int ld, float* klrow, int N, int q, void* stream) {
  if (N == 0) return 0;
  if (!mu || !logvar || !klrow || q < 1 || ld < q) return gp::set_error("gpode_normal_kl_fwd: bad argument");
  return gp::normal_kl_fwd(mu, logvar, ld, klrow, N, q, GP_ST);
}
int gpode_normal_kl_bwd(const float* grow, const float* mu, const float* logvar, int ld, float* gmu, float* glogvar, int ldg, int N, int q,
                        void* stream) {
  if (N == 0) return 0;
  if (!grow || !mu || !logvar || !gmu || !glogvar || q < 1 || ld < q || ldg < q) return gp::set_error("gpode_normal_kl_bwd: bad argument");
  return gp::normal_kl_bwd(grow, mu, logvar, ld, gmu, glogvar, ldg, N, q, GP_ST);
}
int gpode_sigmoid_loglik_splits(size_t rows, size_t inner) { return gp::sigmoid_loglik_splits(rows, inner); }
int gpode_sigmoid_loglik_fwd(const float* X, const float* a, float* z, float* part, size_t rows, size_t inner, size_t nX, int nsplit,
                             void* stream) {
  if (!X || !a || !z || !part) return gp::set_error("gpode_sigmoid_loglik_fwd: null pointer");
  if (!rows || !inner || !nX) return gp::set_error("gpode_sigmoid_loglik_fwd: empty input");
  return gp::sigmoid_loglik_fwd(X, a, z, part, rows, inner, nX, nsplit, GP_ST);
}
int gpode_sigmoid_loglik_bwd(const float* X, const float* z, const float* grow, float* ga, size_t rows, size_t inner, size_t nX,
                             void* stream) {
  if (!X || !z || !grow || !ga) return gp::set_error("gpode_sigmoid_loglik_bwd: null pointer");
  if (!rows || !inner || !nX) return gp::set_error("gpode_sigmoid_loglik_bwd: empty input");
  return gp::sigmoid_loglik_bwd(X, z, grow, ga, rows, inner, nX, GP_ST);
}
int gpode_elbo_all_fwd(const float* lpart, int nl_rows, int nl_values, const float* hs, const float* hv, int N, int q, int M, int Do,
                       const float* Um, const float* Us, float nobs, float* out, void* stream) {
  if (!lpart || !hs || !Um || !Us || !out) return gp::set_error("gpode_elbo_all_fwd: null pointer");
  if (nl_rows < 1 || nl_values < nl_rows || N < 1 || q < 1 || M < 1 || Do < 1) return gp::set_error("gpode_elbo_all_fwd: bad sizes");
  return gp::elbo_all_fwd(lpart, nl_rows, nl_values, hs, hv, N, q, M, Do, Um, Us, nobs, out, GP_ST);
}
int gpode_elbo_all_bwd(const float* g_loss, const float* g_nll, const float* g_kl, const float* g_klu, int nl_rows, const float* hs,
                       const float* hv, int N, int q, int M, int Do, const float* Um, const float* Us, float nobs, float* glrow, float* ghs,
                       float* ghv, float* dUm, float* dUs, void* stream) {
  if (!hs || !Um || !Us || !glrow || !ghs || !dUm || !dUs || (hv && !ghv)) return gp::set_error("gpode_elbo_all_bwd: null pointer");
  if (nl_rows < 1 || N < 1 || q < 1 || M < 1 || Do < 1) return gp::set_error("gpode_elbo_all_bwd: bad sizes");
  return gp::elbo_all_bwd(g_loss, g_nll, g_kl, g_klu, nl_rows, hs, hv, N, q, M, Do, Um, Us, nobs, glrow, ghs, ghv, dUm, dUs, GP_ST);
}
int gpode_elbo_all_bwd_ll(const float* g_loss, const float* g_nll, const float* g_kl, const float* g_klu, int nl_rows, const float* hs,
                          const float* hv, int N, int q, int M, int Do, const float* Um, const float* Us, float nobs, float* glrow, float* ghs,
                          float* ghv, float* dUm, float* dUs, const float* X, const float* z, float* ga, size_t n_logits, size_t nX,
                          void* stream) {
  if (!hs || !Um || !Us || !glrow || !ghs || !dUm || !dUs || (hv && !ghv) || !X || !z || !ga) return gp::set_error("gpode_elbo_all_bwd_ll: null pointer");
  if (nl_rows < 1 || N < 1 || q < 1 || M < 1 || Do < 1 || n_logits < 1 || nX < 1 || n_logits % nX != 0) return gp::set_error("gpode_elbo_all_bwd_ll: bad sizes");
  return gp::elbo_all_bwd_ll(g_loss, g_nll, g_kl, g_klu, nl_rows, hs, hv, N, q, M, Do, Um, Us, nobs, glrow, ghs, ghv, dUm, dUs, X, z, ga, n_logits,
                             nX, GP_ST);
}
int gpode_elbo_fwd(const float* lhood, int nl, const float* klrow, int nk, const float* kl_u, float nobs, float* out, void* stream) {
  if (!lhood || !klrow || !kl_u || !out || nl < 1 || nk < 1) return gp::set_error("gpode_elbo_fwd: bad argument");
  return gp::elbo_fwd(lhood, nl, klrow, nk, kl_u, nobs, out, GP_ST);
}
int gpode_elbo_bwd(const float* gout, int nl, int nk, float nobs, float* glhood, float* gklrow, float* gklu, void* stream) {
  if (!gout || !glhood || !gklrow || !gklu || nl < 1 || nk < 1) return gp::set_error("gpode_elbo_bwd: bad argument");
  return gp::elbo_bwd(gout, nl, nk, nobs, glhood, gklrow, gklu, GP_ST);
}
int gpode_gather_multi(void* grads, const long long* offs, int ntensors, long long total, float* flat, void* stream) {
  if (!grads || !offs || !flat) return gp::set_error("gpode_gather_multi: null pointer");
  return gp::gather_multi((const float* const*)grads, offs, ntensors, total, flat, (hipStream_t)stream);
}

int gpode_adam_multi(void* params, void* grads, void* m1, void* m2, const long long* offs, int ntensors, long long total,
                     float lr, float beta1, float beta2, float eps, int step, int* step_dev, void* stream) {
  if (!params || !grads || !m1 || !m2 || !offs) return gp::set_error("gpode_adam_multi: null pointer");
  return gp::adam_multi((float* const*)params, (const float* const*)grads, (float* const*)m1, (float* const*)m2, offs, ntensors, total,
                        lr, beta1, beta2, eps, step, step_dev, GP_ST);
}
#undef GP_ST

}  // extern "C"
