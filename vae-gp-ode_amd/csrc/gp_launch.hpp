// gp_launch.hpp -- host-side helpers shared by the launchers (error slot, launch checks).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>

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

inline int set_max_lds(const void* kern, size_t bytes) {
  if (bytes <= 64 * 1024) return 0;
  hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) return set_error("hipFuncSetAttribute(max dynamic LDS=%zu): %s", bytes, hipGetErrorString(e));
  return 0;
}

// entry points implemented across the .hip files
int dims_supported(int kernel, int Di, int Do);
int rhs_fwd(int kernel, int Di, int Do, int M, int S, const float* pack, const float* x, int N, float* f, int mode, hipStream_t st);
int rollout_fwd(int kernel, int order, int method, int Di, int Do, int M, int S, const float* pack,
                const float* z0, const float* ts, int N, int T, float* zt, float* xstage, hipStream_t st);
int rollout_bwd(int kernel, int order, int method, int Di, int Do, int M, int S, const float* pack, const float* xstage,
                const float* gzt, const float* ts, int N, int T, float* gz0, float* astage, hipStream_t st);
int rhs_vjp(int kernel, int Di, int Do, int M, int S, const float* pack, const float* x, const float* a, int R, float* gx,
            int prior_only, hipStream_t st);
int param_grad(int kernel, int Di, int Do, int M, int S, const float* pack, const float* xr, const float* ar, int R,
               float* slab, int nchunk, float* gpack, int accumulate, int prior_only, hipStream_t st);
int cache_sizes(int kernel, int Di, int Do, int M, int S, size_t* pack_floats, size_t* ws_floats);
int cache_build_fwd(int kernel, int Di, int Do, int M, int S,
                    const float* raw_ell, const float* raw_var, const float* Z, const float* Um, const float* Us_packed,
                    const float* eps_u, const float* rff_w, const float* rff_eps, const float* rff_u,
                    float* pack, float* ws, float* ell, float* var, float* omega, float* phase, float* u,
                    float* Lu, float* nu, float* u_prior, hipStream_t st);

int kernel_matrix(int kernel, int Di, int Do, const float* raw_ell, const float* raw_var, const float* X, int N,
                  const float* X2, int M2, float* out, hipStream_t st);
int svgp_kl_fwd(int M, int Do, const float* Um, const float* Us, float* kl, hipStream_t st);
int svgp_kl_bwd(int M, int Do, const float* Um, const float* Us, const float* g, float* dUm, float* dUs, hipStream_t st);

int cache_bwd_sizes(int kernel, int Di, int Do, int M, int S, size_t* bws_floats);
int cache_build_bwd(int kernel, int Di, int Do, int M, int S, const float* raw_ell, const float* raw_var, const float* Z,
                    const float* eps_u, const float* pack, const float* ws, float* gpack, float* bws,
                    float* g_raw_ell, float* g_raw_var, float* g_Z, float* g_Um, float* g_Us, hipStream_t st);

}  // namespace gp
