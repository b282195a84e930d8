// bn_sink.hpp -- BatchNorm batch statistics produced by the convolution that WRITES the tensor, not by a pass that reads it again
// (reference: nn.BatchNorm2d in training mode behind every hidden conv layer, experiments/model/core/vae.py:52-60, 107-120).
//
// The convolution kernels of this library are persistent: a workgroup computes many output images.  With a BnSink it also sums
// (y - k) and (y - k)^2 per output channel over everything it stored (k = the layer's running mean before this step: a shift that
// removes the E[y^2] - E[y]^2 cancellation once training has warmed up, and 0 -- plain sums -- at the first step), writes the two
// sums per channel to part[workgroup][C][2], and the LAST workgroup to finish combines all workgroups' sums in a fixed order
// (deterministic, whichever workgroup that is) into mean / invstd / running statistics and the {mean, invstd, gamma, beta} table
// the consuming convolution applies while it stages its input.  Two launches (statistics pass, table kernel) and one read of the
// tensor less per BatchNorm layer.
//
// Visibility without a device-wide fence: the partial sums are stored and loaded with DEVICE-scope (sc1) memory operations, which
// are coherent across the XCDs' L2s by themselves; the ticket is a device-scope atomic.  (__threadfence() would write the whole L2
// back -- the activations this kernel has just stored -- once per workgroup.)
#pragma once
#include <hip/hip_runtime.h>
#include "wave_reduce.hpp"

namespace gp {

struct BnSink {
  float* part;                 // [gridDim.x][C][2] scratch
  unsigned* ticket;            // 0 between launches (the last workgroup resets it)
  const float* gamma;
  const float* beta;
  float* save_mean;
  float* save_invstd;
  float* running_mean;         // may be null (then k = 0 and no running update)
  float* running_var;
  long long* nbt;              // num_batches_tracked, may be null
  float* table;                // [C][4]
  float momentum, eps, count;  // count = elements per channel over the whole batch
};

__device__ __forceinline__ void store_dev(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float load_dev(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ float bn_sink_shift(const BnSink& s, int c) { return s.running_mean ? s.running_mean[c] : 0.f; }

// sm: this workgroup's sums, LDS [C][2] (complete and visible to the workgroup: the caller has synchronised).  All threads call.
template <int C, int NTHR>
__device__ __forceinline__ void bn_sink_publish(const BnSink& s, const float* sm) {
  __shared__ unsigned s_last;
  const int tid = threadIdx.x;
  for (int e = tid; e < 2 * C; e += NTHR) store_dev(s.part + (size_t)blockIdx.x * (2 * C) + e, sm[e]);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");            // this thread's device-scope stores have been performed
  __syncthreads();
  if (tid == 0) {
    const unsigned t = __hip_atomic_fetch_add(s.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = t == gridDim.x - 1;
    if (s_last) __hip_atomic_store(s.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (!s_last) return;
  // every workgroup's sums are in memory.  All threads fetch (many independent device-scope loads in flight): thread (e, sl) adds
  // entry e = (channel, which sum) of the workgroups sl, sl + NSL, ... in order; the NSL slice sums meet in LDS and are added in order
  __shared__ float s_fin[NTHR];
  constexpr int NSL = NTHR / (2 * C);
  static_assert(NSL >= 1 && NSL * 2 * C == NTHR, "threads = slices x 2 C");
  const int nwg = gridDim.x;
  {
    const int e = tid % (2 * C), sl = tid / (2 * C);
    // U independent loads in flight per thread: a device-scope load takes ~1 us, and decnn.1 (C = 64, 512 workgroups, 2 slices)
    // has 256 of them per thread -- four at a time made this tail as long as the convolution itself
    constexpr int U = 16;
    float a[U];
#pragma unroll
    for (int u = 0; u < U; ++u) a[u] = 0.f;
    int g = sl;
    for (; g + (U - 1) * NSL < nwg; g += U * NSL) {
#pragma unroll
      for (int u = 0; u < U; ++u) a[u] += load_dev(s.part + (size_t)(g + u * NSL) * (2 * C) + e);
    }
    for (; g < nwg; g += NSL) a[0] += load_dev(s.part + (size_t)g * (2 * C) + e);
#pragma unroll
    for (int w = 1; w < U; w *= 2)
#pragma unroll
      for (int u = 0; u + w < U; u += 2 * w) a[u] += a[u + w];
    s_fin[tid] = a[0];
  }
  __syncthreads();
  if (tid < C) {
    const int c = tid;
    float out2[2] = {0.f, 0.f};
    for (int sl = 0; sl < NSL; ++sl) { out2[0] += s_fin[sl * 2 * C + 2 * c]; out2[1] += s_fin[sl * 2 * C + 2 * c + 1]; }
    {
      const float k = bn_sink_shift(s, c);
      const float d = out2[0] / s.count;                              // mean - k
      const float m = k + d;
      const float var = fmaxf(out2[1] / s.count - d * d, 0.f);
      const float is = rsqrtf(var + s.eps);
      s.save_mean[c] = m;
      s.save_invstd[c] = is;
      if (s.running_mean) {
        s.running_mean[c] = (1.f - s.momentum) * s.running_mean[c] + s.momentum * m;
        s.running_var[c] = (1.f - s.momentum) * s.running_var[c] + s.momentum * var * (s.count / (s.count - 1.f));
      }
      if (c == 0 && s.nbt) *s.nbt += 1;
      s.table[4 * c + 0] = m; s.table[4 * c + 1] = is; s.table[4 * c + 2] = s.gamma[c]; s.table[4 * c + 3] = s.beta[c];
    }
  }
}

}  // namespace gp
