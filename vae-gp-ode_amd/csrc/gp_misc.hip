// gp_misc.hip -- small operators of the SVGP layer that sit beside the integrator:
//   kernel matrices K(X,X2) for the public kern.K API   (kernels.py:98-110 / :289-303)
//   the whitened inducing KL and its gradient            (svpy.py:144-175)
#include "gp_eval.hpp"
#include "gp_launch.hpp"

namespace gp {

__device__ __forceinline__ float softplus_lower_m(float x) { return (x > 20.f ? x : log1pf(expf(x))) + 1e-12f; }

// RBF: out (Do,N,M2);  grid (ceil(M2/128), N, Do)
__global__ void k_kmat_rbf(int Di, int Do, const float* __restrict__ raw_ell, const float* __restrict__ raw_var,
                           const float* __restrict__ X, int N, const float* __restrict__ X2, int M2, float* __restrict__ out) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = blockIdx.y, d = blockIdx.z;
  if (m >= M2) return;
  float q = 0.f;
  for (int i = 0; i < Di; ++i) {
    float t = (X[n * Di + i] - X2[m * Di + i]) / softplus_lower_m(raw_ell[d * Di + i]);
    q = fmaf(t, t, q);
  }
  out[((size_t)d * N + n) * M2 + m] = softplus_lower_m(raw_var[d]) * expf(-0.5f * q);
}

// DF: out (N*D, M2*D), row (n,a), col (m,b);  grid (ceil(M2*D/128), N*D)
__global__ void k_kmat_df(int D, const float* __restrict__ raw_ell, const float* __restrict__ raw_var,
                          const float* __restrict__ X, int N, const float* __restrict__ X2, int M2, float* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  const int r = blockIdx.y;
  if (c >= M2 * D) return;
  const int n = r / D, a = r % D, m = c / D, b = c % D;
  float r2 = 0.f;
  for (int i = 0; i < D; ++i) { float t = X2[m * D + i] - X[n * D + i]; r2 = fmaf(t, t, r2); }
  float l = softplus_lower_m(raw_ell[a * D + b]);
  float il2 = 1.f / (l * l);
  float da = X2[m * D + a] - X[n * D + a], db = X2[m * D + b] - X[n * D + b];
  float term = da * db * il2 + ((a == b) ? ((float)(D - 1) - r2 * il2) : 0.f);
  out[(size_t)r * (M2 * D) + c] = softplus_lower_m(raw_var[b]) * expf(-0.5f * r2 * il2) * term * il2;
}

int kernel_matrix(int kernel, int Di, int Do, const float* raw_ell, const float* raw_var, const float* X, int N,
                  const float* X2, int M2, float* out, hipStream_t st) {
  if (N <= 0 || M2 <= 0) return 0;
  if (kernel == 0) hipLaunchKernelGGL(k_kmat_rbf, dim3(cdiv(M2, 128), N, Do), 128, 0, st, Di, Do, raw_ell, raw_var, X, N, X2, M2, out);
  else if (kernel == 1) {
    if (Di != Do) return set_error("gpode_kernel_matrix: DF needs Di == Do");
    hipLaunchKernelGGL(k_kmat_df, dim3(cdiv(M2 * Do, 128), N * Do), 128, 0, st, Do, raw_ell, raw_var, X, N, X2, M2, out);
  } else return set_error("gpode_kernel_matrix: kernel %d", kernel);
  return check_launch("kernel_matrix");
}

// ---------------------------------------------------------------------------------------------
// KL(q(u)||N(0,I)) = 0.5 sum_d [ -sum_m log L_mm^2 + sum_m Um[m,d]^2 + ||L_d||_F^2 - M ]
// ONE workgroup of 1024 threads walks every output dimension and reduces in a fixed order: no float atomics (the sum of per-
// dimension atomics made kl_u differ in its last bits from run to run), no memset node in front of it.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_svgp_kl(int M, int Do, const float* __restrict__ Um, const float* __restrict__ Us,
                                                  float* __restrict__ kl) {
  __shared__ float red[16];
  const size_t P = (size_t)M * (M + 1) / 2;
  float acc = 0.f;
  for (size_t e = threadIdx.x; e < P * Do; e += blockDim.x) acc = fmaf(Us[e], Us[e], acc);
  for (int e = threadIdx.x; e < M * Do; e += blockDim.x) {
    const int m = e / Do, d = e % Do;
    const float l = Us[(size_t)d * P + (size_t)m * (m + 1) / 2 + m];
    const float u = Um[e];
    acc += u * u - logf(l * l);
  }
  acc = wave_allreduce_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < 16; ++w) t += red[w];
    *kl = 0.5f * (t - (float)M * (float)Do);
  }
}

// dUm = g Um ; dUs = g Us off the diagonal, g (Us - 1/Us) on it.   g = *gptr (device scalar)
__global__ void k_svgp_kl_bwd(int M, int Do, const float* __restrict__ Um, const float* __restrict__ Us,
                              const float* __restrict__ gptr, float* __restrict__ dUm, float* __restrict__ dUs) {
  const size_t P = (size_t)M * (M + 1) / 2;
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const float g = *gptr;
  if (e < (size_t)M * Do) dUm[e] = g * Um[e];
  if (e < P * Do) {
    const size_t k = e % P;
    // row index n of packed entry k: largest n with n(n+1)/2 <= k
    int n = (int)((sqrtf(8.f * (float)k + 1.f) - 1.f) * 0.5f);
    while ((size_t)(n + 1) * (n + 2) / 2 <= k) ++n;
    while ((size_t)n * (n + 1) / 2 > k) --n;
    const bool diag = (k - (size_t)n * (n + 1) / 2) == (size_t)n;
    const float v = Us[e];
    dUs[e] = g * (diag ? v - 1.f / v : v);
  }
}

int svgp_kl_fwd(int M, int Do, const float* Um, const float* Us, float* kl, hipStream_t st) {
  hipLaunchKernelGGL(k_svgp_kl, 1, 1024, 0, st, M, Do, Um, Us, kl);
  return check_launch("svgp_kl");
}

int svgp_kl_bwd(int M, int Do, const float* Um, const float* Us, const float* g, float* dUm, float* dUs, hipStream_t st) {
  const size_t P = (size_t)M * (M + 1) / 2 * Do;
  hipLaunchKernelGGL(k_svgp_kl_bwd, (unsigned)((P + 255) / 256), 256, 0, st, M, Do, Um, Us, g, dUm, dUs);
  return check_launch("svgp_kl_bwd");
}

}  // namespace gp
