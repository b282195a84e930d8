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

// ---------------------------------------------------------------------------------------------
// Device-side noise of a whole step in ONE launch (model/core/noise.py: DeviceNoise).  The reference draws with numpy on the host
// (kernels.py:13-26,134-137; svpy.py:12-27,94) and torch.randn_like (vae.py:76); throughput runs draw on the device instead:
// out[0 .. n_normal) ~ N(0,1), out[n_normal .. n_normal + n_uniform) ~ U[0,1), from Philox4x32-10 keyed by `seed`, counter =
// (element quad, draw number).  The draw number lives in device memory (state[0]) and is advanced by the LAST workgroup to
// finish (ticket state[1]; every workgroup has read state[0] before it takes its ticket), so a captured launch produces fresh
// numbers at every replay without any host-side bookkeeping, and the same (seed, draw number) gives the same numbers on every rank.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__global__ __launch_bounds__(256) void k_noise_fill(float* __restrict__ out, long long n_normal, long long n_uniform,
                                                     unsigned long long seed, unsigned long long* __restrict__ state) {
  const unsigned long long draw = state[0];
  const long long qn = (n_normal + 3) / 4, qu = (n_uniform + 3) / 4;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t < qn + qu) {
    unsigned r[4];
    philox4x32_10((unsigned)t, (unsigned)((unsigned long long)t >> 32), (unsigned)draw, (unsigned)(draw >> 32), (unsigned)seed,
                  (unsigned)(seed >> 32), r);
    float v[4];
    const bool normal = t < qn;
    if (normal) {                                    // Box-Muller on two pairs; u in (0, 1]: 2^-33 .. 1, radius <= 6.76
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const float u = fmaf((float)r[2 * p], 2.3283064365386963e-10f, 1.1641532182693481e-10f);
        const float rad = sqrtf(-2.f * logf(u));
        const float ang = 6.283185307179586f * ((float)(r[2 * p + 1] >> 8) * 5.9604644775390625e-08f);
        float sn, cs;
        sincosf(ang, &sn, &cs);
        v[2 * p] = rad * cs;
        v[2 * p + 1] = rad * sn;
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = (float)(r[k] >> 8) * 5.9604644775390625e-08f;   // 24 bits: [0, 1)
    }
    const long long e0 = normal ? 4 * t : n_normal + 4 * (t - qn), e1 = normal ? n_normal : n_normal + n_uniform;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (e0 + k < e1) out[e0 + k] = v[k];
  }
  __syncthreads();
  if (threadIdx.x == 0) {                            // only the counter passes between workgroups: a device-scope atomic, no fence
    const unsigned long long ticket = atomicAdd(&state[1], 1ull);
    if (ticket == (unsigned long long)gridDim.x - 1) {
      state[1] = 0;
      state[0] = draw + 1;
    }
  }
}

int noise_fill(float* out, long long n_normal, long long n_uniform, unsigned long long seed, unsigned long long* state, hipStream_t st) {
  const long long threads = (n_normal + 3) / 4 + (n_uniform + 3) / 4;
  if (threads <= 0) return 0;
  hipLaunchKernelGGL(k_noise_fill, (unsigned)((threads + 255) / 256), 256, 0, st, out, n_normal, n_uniform, seed, state);
  return check_launch("noise_fill");
}

}  // namespace gp
