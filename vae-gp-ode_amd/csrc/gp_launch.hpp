// gp_launch.hpp -- host-side helpers shared by the launchers (error slot, launch checks).
#pragma once
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <mutex>
#include <unordered_map>

namespace gp {

char* error_slot();  // thread-local 512-byte buffer (capi.hip)

inline int set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(error_slot(), 512, fmt, ap);
  va_end(ap);
  return 1;
}

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error("%s: launch failed: %s", what, hipGetErrorString(e));
  return 0;
}

// Raise a kernel's dynamic-LDS limit above the 64 KB default.  The attribute is set once per (kernel, size) and
// remembered: hipFuncSetAttribute is not a stream operation (it is refused while the stream is being captured into
// a graph), and the first eager run of a step has already set every attribute its replay needs.
inline int set_max_lds(const void* kern, size_t bytes) {
  if (bytes <= 64 * 1024) return 0;
  static std::mutex mu;
  static std::unordered_map<const void*, size_t> done;
  std::lock_guard<std::mutex> lock(mu);
  auto it = done.find(kern);
  if (it != done.end() && it->second >= bytes) return 0;
  hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) return set_error("hipFuncSetAttribute(max dynamic LDS=%zu): %s", bytes, hipGetErrorString(e));
  done[kern] = bytes;
  return 0;
}

// Compute units of the current device: the persistent kernels launch one (or two) workgroup(s) per CU.  Queried once (the first
// eager run of a step, never inside a capture); clamped to 256 because the split-sum scratch buffers are sized for that.
inline int num_cus() {
  static const int n = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
      return 256;
    return v < 256 ? v : 256;
  }();
  return n;
}

// Monte-Carlo draws batched into ONE launch (the reference loops `for l in range(L)` over whole flow calls, odegpvae.py:41-43):
// blockIdx.y = draw.  Strides in floats of the operands that differ per draw -- 0 = shared by all draws (the inducing locations
// handed to the prior-only rhs, the initial states z0 of a rollout).  Which operand each stride belongs to is listed per entry point.
struct Draws {
  int nd = 1;
  size_t pack = 0, in = 0, out = 0, in2 = 0, out2 = 0;
};

// entry points implemented across the .hip files
int dims_supported(int kernel, int Di, int Do);
//   rhs_fwd:     in = x, out = f                       rollout_fwd: in = z0, out = zt, out2 = xstage
//   rollout_bwd: in = xstage, in2 = gzt, out = gz0, out2 = astage
//   rhs_vjp:     in = x, in2 = a, out = gx             param_grad:  in = xr, in2 = ar, out = gpack (slab: nchunk * pack_floats per draw)
int rhs_fwd(int kernel, int Di, int Do, int M, int S, const float* pack, const float* x, int N, float* f, int mode, hipStream_t st,
            Draws dw = Draws{});
int rollout_fwd(int kernel, int order, int method, int Di, int Do, int M, int S, const float* pack,
                const float* z0, const float* ts, int N, int T, float* zt, float* xstage, hipStream_t st, Draws dw = Draws{});
int rollout_bwd(int kernel, int order, int method, int Di, int Do, int M, int S, const float* pack, const float* xstage,
                const float* gzt, const float* ts, int N, int T, float* gz0, float* astage, hipStream_t st, Draws dw = Draws{});
int rollout_bwd_pgrad_chunks(int kernel, int order, int method, int Di, int Do, int M, int S, int N);
int rollout_bwd_pgrad(int kernel, int order, int method, int Di, int Do, int M, int S, const float* pack, const float* xstage,
                      const float* gzt, const float* ts, int N, int T, float* gz0, float* astage, float* slab, int nchunk,
                      float* gpack, hipStream_t st, Draws dw = Draws{});
int rhs_vjp(int kernel, int Di, int Do, int M, int S, const float* pack, const float* x, const float* a, int R, float* gx,
            int prior_only, hipStream_t st, Draws dw = Draws{});
int param_grad(int kernel, int Di, int Do, int M, int S, const float* pack, const float* xr, const float* ar, int R,
               float* slab, int nchunk, float* gpack, int accumulate, int prior_only, hipStream_t st, Draws dw = Draws{});
int cache_sizes(int kernel, int Di, int Do, int M, int S, size_t* pack_floats, size_t* ws_floats, int nd = 1);
int cache_build_fwd(int kernel, int Di, int Do, int M, int S, int nd,
                    const float* raw_ell, const float* raw_var, const float* Z, const float* Um, const float* Us_packed,
                    const float* eps_u, const float* rff_w, const float* rff_eps, const float* rff_u,
                    float* pack, float* ws, float* ell, float* var, float* omega, float* phase, float* u,
                    float* Lu, float* nu, float* u_prior, hipStream_t st);

size_t kern_scratch_floats(int kernel, int Di, int Do, int M, int S);
int kern_cache(int kernel, int Di, int Do, int S, const float* raw_ell, const float* raw_var, const float* rff_w, const float* rff_eps,
               const float* rff_u, float* pack, float* omega, float* phase, hipStream_t st);
int compute_nu_ws(int kernel, int Di, int Do, int M, size_t* ws_floats);
int compute_nu(int kernel, int Di, int Do, int M, const float* Ku, const float* u_prior, const float* u, float* nu, float* ws, hipStream_t st);
int f_update(int kernel, int Di, int Do, int M, const float* raw_ell, const float* raw_var, const float* x2, const float* nu,
             const float* x, int N, float* out, float* pack, hipStream_t st);
int kernel_matrix(int kernel, int Di, int Do, const float* raw_ell, const float* raw_var, const float* X, int N,
                  const float* X2, int M2, float* out, hipStream_t st);
int conditional_ws(int Di, int Do, int M, int N, size_t* ws_floats);
int conditional(int Di, int Do, int M, int N, const float* raw_ell, const float* raw_var, const float* Z, const float* Um,
                const float* Us_packed, int us_rank1, const float* x, int full_cov, float* mean, float* var, float* ws, hipStream_t st);
int svgp_kl_fwd(int M, int Do, const float* Um, const float* Us, float* kl, hipStream_t st);
int noise_fill(float* out, long long n_normal, long long n_uniform, unsigned long long seed, unsigned long long* state, hipStream_t st);
int svgp_kl_bwd(int M, int Do, const float* Um, const float* Us, const float* g, float* dUm, float* dUs, hipStream_t st);

void set_backward_solves(int mode);
int get_backward_solves();
int cache_bwd_sizes(int kernel, int Di, int Do, int M, int S, int nd, size_t* bws_floats);
int cache_build_bwd(int kernel, int Di, int Do, int M, int S, int nd, const float* raw_ell, const float* raw_var, const float* Z,
                    const float* eps_u, const float* pack, const float* ws, float* gpack, float* bws,
                    float* g_raw_ell, float* g_raw_var, float* g_Z, float* g_Um, float* g_Us, int prepared, hipStream_t st);
int cache_bwd_prepare(int kernel, int Di, int Do, int M, int S, int nd, const float* ws, float* bws, hipStream_t st);

// conv VAE blocks (vae_conv.hip)
int conv2d_fwd(const float* x, const float* w, const float* bias, float* y, int B, int Ci, int H, int W, int Co, int K, int S,
               int P, int Ho, int Wo, hipStream_t st, size_t xbs = 0);
size_t convT_fwd_stats_scratch(int Co_out);
int convT_fwd_stats(const float* x, const float* x_bn, const float* w, const float* bias, float* y, int B, int Ci, int H, int W, int Co, int K,
                    int S, int P, int Ho, int Wo, const float* gamma, const float* beta, float* save_mean, float* save_invstd,
                    float* running_mean, float* running_var, long long* nbt, float momentum, float eps, float* table, float* scratch,
                    int slot, hipStream_t st);
int conv2d_bwd_data(const float* gy, const float* w, const float* bias, float* gx, int B, int Ci, int H, int W, int Co, int K, int S,
                    int P, int Ho, int Wo, const float* gy_bn, hipStream_t st);
size_t conv_wgrad_scratch(int B, int Ci, int Co, int K);
int conv2d_bwd_weight(const float* x, const float* gy, float* gw, float* gbias, float* scratch, int B, int Ci, int H, int W, int Co,
                      int K, int S, int P, int Ho, int Wo, const float* gy_bn, hipStream_t st, size_t xbs = 0);
size_t bn_scratch(int B, int C);
int bn_fwd(const float* x, const float* gamma, const float* beta, float* y, float* save_mean, float* save_invstd,
           float* running_mean, float* running_var, long long* num_batches_tracked, float momentum, float eps, int B, int C, int HW,
           int relu, float* scratch, hipStream_t st);
int bn_bwd(const float* x, const float* gy, const float* gamma, const float* beta, const float* save_mean, const float* save_invstd,
           float* gx, float* ggamma, float* gbeta, float* gx_chansum, int B, int C, int HW, int relu, float* scratch, hipStream_t st);
int bn_stats(const float* x, const float* gamma, const float* beta, float* save_mean, float* save_invstd, float* running_mean,
             float* running_var, long long* num_batches_tracked, float momentum, float eps, float* table, int B, int C, int HW,
             float* scratch, hipStream_t st);
int bn_moments(const float* x, float* mom, int B, int C, int HW, float* scratch, hipStream_t st);
int bn_finalize(const float* gathered, int W, const float* gamma, const float* beta, float* save_mean, float* save_invstd,
                float* running_mean, float* running_var, long long* num_batches_tracked, float momentum, float eps, float* table, int C,
                hipStream_t st);
int bn_apply(const float* x, const float* table, float* y, int B, int C, int HW, int relu, hipStream_t st);
// decnn.10's input gradient fused with the BatchNorm + ReLU backward in front of it (vae_conv_tiled.hip)
int dec10_bn_scratch_floats();
int dec10_bn_bwd_sums(const float* c, const float* gy, const float* w, const float* gamma, const float* beta, const float* mean,
                      const float* invstd, float* sums, int B, float* scratch, hipStream_t st, float* gw = nullptr, float* wscratch = nullptr, float* gbias = nullptr);
int dec10_bn_wgrad_scratch_floats();
int dec10_bn_bwd_apply(const float* c, const float* gy, const float* w, const float* gamma, const float* beta, const float* mean,
                       const float* invstd, const float* gathered, const float* wts, int W, float count_all, float* gc, float* ggamma,
                       float* gbeta, float* gc_chansum, int B, float* scratch, hipStream_t st);
// final reductions of per-workgroup partials, recorded between defer_reductions(1) / (0) and run together by flush_reductions
// (vae_norm.hip); outside such a bracket reduce_job launches at once
struct RedJob { const float* part; float* out; int nsplit, n, kind, a, b, c; };
int reduce_job(const RedJob& job, hipStream_t st);
int reduce_jobs(const RedJob* jobs, int nj, hipStream_t st);
void defer_reductions(int mode);
int flush_reductions(hipStream_t st);
int bn_bwd_sums(const float* x, const float* gy, const float* gamma, const float* beta, const float* save_mean, const float* save_invstd,
                float* sums, int B, int C, int HW, int relu, float* scratch, hipStream_t st);
int bn_bwd_apply(const float* x, const float* gy, const float* gamma, const float* beta, const float* save_mean, const float* save_invstd,
                 const float* gathered, const float* wts, int W, float count, float* gx, float* ggamma, float* gbeta, float* gx_chansum, int B, int C, int HW, int relu,
                 float* scratch, hipStream_t st);
int bn_eval(const float* x, const float* gy, const float* gamma, const float* beta, const float* running_mean, const float* running_var,
            float eps, float* out, int B, int C, int HW, int relu, hipStream_t st);
int chan_sum(const float* v, float* out, int B, int C, int HW, float* scratch, hipStream_t st);
int act_fwd(const float* x, float* y, size_t n, int mode, hipStream_t st);
int act_bwd(const float* y, const float* gy, float* gx, size_t n, int mode, hipStream_t st);
int linear_fwd(const float* x, const float* w, const float* bias, float* y, int B, int In, int Out, hipStream_t st);
int linear_bwd(const float* x, const float* w, const float* gy, float* gx, float* gw, float* gb, int B, int In, int Out, float* scratch, hipStream_t st);
size_t linear_bwd_scratch(int B, int In, int Out);
int linear_relu_fwd(const float* x, const float* w, const float* bias, float* y, int B, int In, int Out, hipStream_t st);
int linear_relu_bwd(const float* x, const float* w, const float* gy, float* gx, float* gw, float* gb, int B, int In, int Out, hipStream_t st);
int loglik_fwd(const float* X, const float* z, float* ll, size_t n, size_t nX, hipStream_t st);
int loglik_bwd(const float* X, const float* z, const float* g, float* gz, size_t n, size_t nX, hipStream_t st);
int loglik_rowsum_fwd(const float* X, const float* z, float* out, size_t rows, size_t inner, size_t nX, hipStream_t st);
int loglik_rowsum_bwd(const float* X, const float* z, const float* grow, float* gz, size_t rows, size_t inner, size_t nX, hipStream_t st);
int elbo_all_bwd_ll(const float* g0, const float* g1, const float* g2, const float* g3, int nl_rows, const float* hs, const float* hv, int N,
                    int q, int M, int Do, const float* Um, const float* Us, float nobs, float* glrow, float* ghs, float* ghv, float* dUm,
                    float* dUs, const float* X, const float* z, float* ga, size_t n, size_t nX, hipStream_t st);
int reparam_kl_fwd(const float* mu, const float* logvar, int ld, const float* eps, float* z, float* klpart, int N, int q, hipStream_t st);
int reparam_kl_bwd(const float* gz, const float* gklpart, const float* mu, const float* logvar, int ld, const float* eps, float* gmu,
                   float* glogvar, int ldg, int N, int q, hipStream_t st);
int elbo_all_fwd_kl(const float* lpart, int nl_rows, int nl_values, const float* kls, int nks, const float* klv, int nkv, int N, int M, int Do,
                    const float* Um, const float* Us, float nobs, float* out, hipStream_t st);
int elbo_all_bwd_ll_kl(const float* g0, const float* g1, const float* g2, const float* g3, int nl_rows, int N, int M, int Do, const float* Um,
                       const float* Us, float nobs, float* glrow, float* gkls, int nks, float* gklv, int nkv, float* dUm, float* dUs,
                       const float* X, const float* z, float* ga, size_t n, size_t nX, hipStream_t st);
int reparam_fwd(const float* mu, const float* logvar, int ld, const float* eps, float* z, int N, int q, hipStream_t st);
int reparam_bwd(const float* gz, const float* logvar, int ld, const float* eps, float* gmu, float* glogvar, int ldg, int N, int q, hipStream_t st);
int normal_kl_fwd(const float* mu, const float* logvar, int ld, float* klrow, int N, int q, hipStream_t st);
int normal_kl_bwd(const float* grow, const float* mu, const float* logvar, int ld, float* gmu, float* glogvar, int ldg, int N, int q, hipStream_t st);
int sigmoid_loglik_splits(size_t rows, size_t inner);
int sigmoid_loglik_fwd(const float* X, const float* a, float* z, float* part, size_t rows, size_t inner, size_t nX, int nsplit, hipStream_t st);
int sigmoid_loglik_bwd(const float* X, const float* z, const float* grow, float* ga, size_t rows, size_t inner, size_t nX, hipStream_t st);
int elbo_all_fwd(const float* lpart, int nl_rows, int nl_values, const float* hs, const float* hv, int N, int q, int M, int Do,
                 const float* Um, const float* Us, float nobs, float* out, hipStream_t st);
int elbo_all_bwd(const float* g0, const float* g1, const float* g2, const float* g3, int nl_rows, const float* hs, const float* hv, int N,
                 int q, int M, int Do, const float* Um, const float* Us, float nobs, float* glrow, float* ghs, float* ghv, float* dUm,
                 float* dUs, hipStream_t st);
int elbo_fwd(const float* lhood, int nl, const float* klrow, int nk, const float* kl_u, float nobs, float* out, hipStream_t st);
int elbo_bwd(const float* gout, int nl, int nk, float nobs, float* glhood, float* gklrow, float* gklu, hipStream_t st);
int gather_multi(const float* const* grads, const long long* offs, int ntensors, long long total, float* flat, hipStream_t st);
int adam_multi(float* const* params, const float* const* grads, float* const* m1, float* const* m2, const long long* offs,
               int ntensors, long long total, float lr, float beta1, float beta2, float eps, int step, int* step_dev, hipStream_t st);

// big factors (np a multiple of 128, >= 1024) take the panelled / matrix-core kernels; GPODE_SMALL_FACTOR_KERNELS=1 keeps the
// 32-tile kernels for every size (an A/B switch for tests and profiling, same results up to summation order)
inline bool big_factor(int np) {
  static const bool off = [] { const char* e = getenv("GPODE_SMALL_FACTOR_KERNELS"); return e && e[0] == '1'; }();
  return !off && np % 128 == 0 && np >= 1024;
}

}  // namespace gp
