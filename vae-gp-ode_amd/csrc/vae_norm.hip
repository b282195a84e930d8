// vae_norm.hip -- BatchNorm2d (training mode) and per-channel sums for the VAE encoder / decoder
// (reference: experiments/model/core/vae.py:52-60 and 64-84, nn.BatchNorm2d after every hidden conv layer).
//
// All kernels are HBM-bound passes over an activation tensor (B, C, HW).  Layout of the work: grid (C, nsplit);
// a workgroup owns channel c and a contiguous slab of images, so the channel's scalars are workgroup constants and
// the loop body has no integer division (float4 accesses when HW % 4 == 0).  Pass counts:
//   forward   statistics (1 read)  + normalise / affine / ReLU (1 read, 1 write; finalises the statistics itself)
//   backward  channel sums (2 reads: x, gy) + apply (2 reads, 1 write); the ReLU mask is recomputed from x with the
//             same pinned arithmetic as the forward pass, so the forward output y is not read again.
// Variance: sums of (x - k) and (x - k)^2 with a per-channel shift k (mean of the first <= 64 elements of the
// channel), so the E[x^2] - E[x]^2 cancellation is relative to (mean - k)^2 / var = O(1), not mean^2 / var.
// Reductions use fixed slabs and a fixed combine order: results are bitwise reproducible.
#include <hip/hip_runtime.h>
#include <mutex>
#include "gp_launch.hpp"
#include "wave_reduce.hpp"
#include "bn_math.hpp"

namespace gp {

namespace {

// visit channel c of images [b0, b0 + nb): f4(offset of 4 consecutive floats) when HW % 4 == 0, else f1(offset)
template <class F4, class F1>
__device__ __forceinline__ void chan_slab(int C, int HW, int c, int b0, int nb, F4&& f4, F1&& f1) {
  if ((HW & 3) == 0) {
    const int hw4 = HW >> 2, work = nb * hw4;
    int q = threadIdx.x, b = q / hw4, p = q - b * hw4;
    const int sb = 256 / hw4, sp = 256 - sb * hw4;
#pragma unroll 4
    for (; q < work; q += 256) {
      f4(((size_t)(b0 + b) * C + c) * HW + 4 * p);
      p += sp; b += sb;
      if (p >= hw4) { p -= hw4; ++b; }
    }
  } else {
    const int work = nb * HW;
    int q = threadIdx.x, b = q / HW, p = q - b * HW;
    const int sb = 256 / HW, sp = 256 - sb * HW;
#pragma unroll 4
    for (; q < work; q += 256) {
      f1(((size_t)(b0 + b) * C + c) * HW + p);
      p += sp; b += sb;
      if (p >= HW) { p -= HW; ++b; }
    }
  }
}

// workgroup sum of NV per-thread values -> part[0..NV) (written by thread 0..NV-1); 256 threads
template <int NV> __device__ __forceinline__ void block_sum_store(const float (&v)[NV], float* __restrict__ dst) {
  __shared__ float red[4][NV];
  float out[NV];
  wave_sum_multi<NV>(v, out);
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) red[threadIdx.x >> 6][i] = out[i];
  }
  __syncthreads();
  if (threadIdx.x < NV) dst[threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// sum over the splits (<= 64) of part[s][c][0..1]: lane s loads split s; identical in every wavefront of every workgroup
__device__ __forceinline__ void split_sums(const float* __restrict__ part, int nsplit, int C, int c, float& a, float& b) {
  const int lane = threadIdx.x & 63;
  const float in2[2] = {lane < nsplit ? part[((size_t)lane * C + c) * 2] : 0.f, lane < nsplit ? part[((size_t)lane * C + c) * 2 + 1] : 0.f};
  float out2[2];
  wave_sum_multi<2>(in2, out2);
  a = out2[0];
  b = out2[1];
}

// shift k(c): mean of the first min(HW, 64) elements of image 0 (every wavefront of every workgroup computes the same value)
__device__ __forceinline__ float chan_shift(const float* __restrict__ x, int HW, int c) {
  const int n = HW < 64 ? HW : 64, lane = threadIdx.x & 63;
  const float in1[1] = {lane < n ? x[(size_t)c * HW + lane] : 0.f};
  float out1[1];
  wave_sum_multi<1>(in1, out1);
  return out1[0] / (float)n;
}

// part[split][c] = {sum (x - k), sum (x - k)^2}; shift[c] = k
__global__ __launch_bounds__(256) void k_bn_stats(const float* __restrict__ x, int B, int C, int HW, int bps, float* __restrict__ part,
                                                   float* __restrict__ shift) {
  const int c = blockIdx.x, b0 = blockIdx.y * bps, nb = min(B, b0 + bps) - b0;
  const float k = chan_shift(x, HW, c);
  float s0 = 0.f, s1 = 0.f, t0 = 0.f, t1 = 0.f;      // two interleaved accumulator pairs
  chan_slab(C, HW, c, b0, nb,
            [&](size_t i) {
              const float4 v = *reinterpret_cast<const float4*>(x + i);
              const float a = v.x - k, b = v.y - k, d = v.z - k, e = v.w - k;
              s0 += a; t0 += b; s0 += d; t0 += e;
              s1 = fmaf(a, a, s1); t1 = fmaf(b, b, t1); s1 = fmaf(d, d, s1); t1 = fmaf(e, e, t1);
            },
            [&](size_t i) {
              const float a = x[i] - k;
              s0 += a;
              s1 = fmaf(a, a, s1);
            });
  const float in2[2] = {s0 + t0, s1 + t1};
  block_sum_store<2>(in2, part + ((size_t)blockIdx.y * C + c) * 2);
  if (blockIdx.y == 0 && threadIdx.x == 0) shift[c] = k;
}

// batch statistics from the split sums (biased variance for the normalisation, unbiased for the running estimate,
// momentum as nn.BatchNorm2d), then normalise / affine / ReLU.  Every workgroup of channel c derives the same statistics;
// split 0 publishes them.
__global__ __launch_bounds__(256) void k_bn_apply(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                   const float* __restrict__ part, const float* __restrict__ shift, int nsplit, float count,
                                                   float eps, float momentum, float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                   float* __restrict__ running_mean, float* __restrict__ running_var,
                                                   long long* __restrict__ num_batches_tracked, float* __restrict__ y,
                                                   int B, int C, int HW, int bps, int relu) {
  const int c = blockIdx.x, b0 = blockIdx.y * bps, nb = min(B, b0 + bps) - b0;
  float s0, s1;
  split_sums(part, nsplit, C, c, s0, s1);
  const float d = s0 / count;                        // mean - k
  const float m = shift[c] + d;
  const float var = fmaxf(s1 / count - d * d, 0.f);
  const float is = rsqrtf(var + eps);
  if (blockIdx.y == 0 && threadIdx.x == 0) {
    save_mean[c] = m;
    save_invstd[c] = is;
    if (running_mean) {
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * m;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * var * (count / (count - 1.f));
    }
    if (c == 0 && num_batches_tracked) *num_batches_tracked += 1;
  }
  const float g = gamma[c], bt = beta[c];
  const float lo = relu ? 0.f : -INFINITY;
  chan_slab(C, HW, c, b0, nb,
            [&](size_t i) {
              const float4 v = *reinterpret_cast<const float4*>(x + i);
              float4 o;
              o.x = fmaxf(bn_affine(v.x, m, is, g, bt), lo); o.y = fmaxf(bn_affine(v.y, m, is, g, bt), lo);
              o.z = fmaxf(bn_affine(v.z, m, is, g, bt), lo); o.w = fmaxf(bn_affine(v.w, m, is, g, bt), lo);
              *reinterpret_cast<float4*>(y + i) = o;
            },
            [&](size_t i) { y[i] = fmaxf(bn_affine(x[i], m, is, g, bt), lo); });
}

// statistics only (no output): finalises the split sums into save_mean / save_invstd / running statistics and writes the
// per-channel table {mean, invstd, gamma, beta} that a consuming convolution applies on the fly (conv_mfma.hpp, in_bn)
__global__ __launch_bounds__(64) void k_bn_table(const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ part,
                                                 const float* __restrict__ shift, int nsplit, int C, float count, float eps, float momentum,
                                                 float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                 float* __restrict__ running_mean, float* __restrict__ running_var,
                                                 long long* __restrict__ num_batches_tracked, float* __restrict__ table) {
  const int c = blockIdx.x;
  float s0, s1;
  split_sums(part, nsplit, C, c, s0, s1);
  if (threadIdx.x != 0) return;
  const float d = s0 / count;
  const float m = shift[c] + d;
  const float var = fmaxf(s1 / count - d * d, 0.f);
  const float is = rsqrtf(var + eps);
  save_mean[c] = m;
  save_invstd[c] = is;
  if (running_mean) {
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * m;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * var * (count / (count - 1.f));
  }
  if (c == 0 && num_batches_tracked) *num_batches_tracked += 1;
  table[4 * c + 0] = m; table[4 * c + 1] = is; table[4 * c + 2] = gamma[c]; table[4 * c + 3] = beta[c];
}

// g = gy masked by the ReLU; part[split][c] = {sum g, sum g * xhat}
__global__ __launch_bounds__(256) void k_bn_bwd_sums(const float* __restrict__ x, const float* __restrict__ gy, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, const float* __restrict__ mean,
                                                      const float* __restrict__ invstd, int B, int C, int HW, int bps, int relu,
                                                      float* __restrict__ part) {
  const int c = blockIdx.x, b0 = blockIdx.y * bps, nb = min(B, b0 + bps) - b0;
  const float m = mean[c], is = invstd[c], g = gamma[c], bt = beta[c];
  float s0 = 0.f, s1 = 0.f, t0 = 0.f, t1 = 0.f;
  auto one = [&](float xv, float gv, float& a0, float& a1) {
    const float xh = bn_xhat(xv, m, is);
    if (relu && !(__fmaf_rn(xh, g, bt) > 0.f)) gv = 0.f;
    a0 += gv;
    a1 = fmaf(gv, xh, a1);
  };
  chan_slab(C, HW, c, b0, nb,
            [&](size_t i) {
              const float4 v = *reinterpret_cast<const float4*>(x + i), w = *reinterpret_cast<const float4*>(gy + i);
              one(v.x, w.x, s0, s1); one(v.y, w.y, t0, t1); one(v.z, w.z, s0, s1); one(v.w, w.w, t0, t1);
            },
            [&](size_t i) { one(x[i], gy[i], s0, s1); });
  const float in2[2] = {s0 + t0, s1 + t1};
  block_sum_store<2>(in2, part + ((size_t)blockIdx.y * C + c) * 2);
}

__global__ __launch_bounds__(256) void k_bn_bwd_apply(const float* __restrict__ x, const float* __restrict__ gy, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float* __restrict__ mean,
                                                       const float* __restrict__ invstd, const float* __restrict__ part, int nsplit,
                                                       float count, float* __restrict__ ggamma, float* __restrict__ gbeta,
                                                       float* __restrict__ gx, float* __restrict__ part_gx, int B, int C, int HW, int bps,
                                                       int relu, const float* __restrict__ gathered, const float* __restrict__ wts, int W) {
  const int c = blockIdx.x, b0 = blockIdx.y * bps, nb = min(B, b0 + bps) - b0;
  const float m = mean[c], is = invstd[c], g = gamma[c], bt = beta[c];
  float sa, sb;                                      // sum g, sum g * xhat over all splits: also the affine gradients
  split_sums(part, nsplit, C, c, sa, sb);
  if (blockIdx.y == 0 && threadIdx.x == 0) { gbeta[c] = sa; ggamma[c] = sb; }
  // cross-rank statistics: the centring terms come from the (weighted) sums over ALL ranks and the global element count; the
  // affine gradients above stay this rank's own sums (the gradient all-reduce combines them like every other parameter)
  if (gathered) {                                    // gathered[r][2C]: rank r's sums; wts[r] = its weight relative to this rank
    const int lane = threadIdx.x & 63;
    const float wr = lane < W ? wts[lane] : 0.f;
    const float in2[2] = {lane < W ? wr * gathered[(size_t)lane * 2 * C + 2 * c] : 0.f, lane < W ? wr * gathered[(size_t)lane * 2 * C + 2 * c + 1] : 0.f};
    float out2[2];
    wave_sum_multi<2>(in2, out2);
    sa = out2[0];
    sb = out2[1];
  }
  const float ic = 1.f / count, ca = sa * ic, cb = sb * ic, sc = g * is;
  auto one = [&](float xv, float gv) {
    const float xh = bn_xhat(xv, m, is);
    if (relu && !(__fmaf_rn(xh, g, bt) > 0.f)) gv = 0.f;
    return sc * (gv - ca - xh * cb);
  };
  float s0 = 0.f, t0 = 0.f;                          // channel sum of gx = the bias gradient of the layer that produced x
  chan_slab(C, HW, c, b0, nb,
            [&](size_t i) {
              const float4 v = *reinterpret_cast<const float4*>(x + i), w = *reinterpret_cast<const float4*>(gy + i);
              float4 o;
              o.x = one(v.x, w.x); o.y = one(v.y, w.y); o.z = one(v.z, w.z); o.w = one(v.w, w.w);
              *reinterpret_cast<float4*>(gx + i) = o;
              s0 += o.x; t0 += o.y; s0 += o.z; t0 += o.w;
            },
            [&](size_t i) {
              const float o = one(x[i], gy[i]);
              gx[i] = o;
              s0 += o;
            });
  if (part_gx) {
    const float in2[2] = {s0 + t0, 0.f};
    block_sum_store<2>(in2, part_gx + ((size_t)blockIdx.y * C + c) * 2);
  }
}

// ---- small batches: ONE workgroup per channel walks all images twice (statistics / sums, then apply) -- a launch instead of two
// where the tensor is a few hundred KB and each launch is a graph node of ~5 us on the step's critical chain (configs[0]) ----------
__device__ __forceinline__ void block_sum2(float a, float b, float& sa, float& sb) {
  __shared__ float red2[4][2];
  const float in2[2] = {a, b};
  float out2[2];
  wave_sum_multi<2>(in2, out2);
  __syncthreads();                                   // red2 may still be read from a previous call
  if ((threadIdx.x & 63) == 0) { red2[threadIdx.x >> 6][0] = out2[0]; red2[threadIdx.x >> 6][1] = out2[1]; }
  __syncthreads();
  sa = (red2[0][0] + red2[1][0]) + (red2[2][0] + red2[3][0]);
  sb = (red2[0][1] + red2[1][1]) + (red2[2][1] + red2[3][1]);
}

// y == nullptr: statistics + table only (the consumer applies the normalisation itself)
__global__ __launch_bounds__(256) void k_bn_fwd_small(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float count, float eps, float momentum, float* __restrict__ save_mean,
                                                       float* __restrict__ save_invstd, float* __restrict__ running_mean,
                                                       float* __restrict__ running_var, long long* __restrict__ num_batches_tracked,
                                                       float* __restrict__ y, float* __restrict__ table, int B, int C, int HW, int relu) {
  const int c = blockIdx.x;
  const float k = chan_shift(x, HW, c);
  float s0 = 0.f, s1 = 0.f, t0 = 0.f, t1 = 0.f;
  chan_slab(C, HW, c, 0, B,
            [&](size_t i) {
              const float4 v = *reinterpret_cast<const float4*>(x + i);
              const float a = v.x - k, b = v.y - k, d = v.z - k, e = v.w - k;
              s0 += a; t0 += b; s0 += d; t0 += e;
              s1 = fmaf(a, a, s1); t1 = fmaf(b, b, t1); s1 = fmaf(d, d, s1); t1 = fmaf(e, e, t1);
            },
            [&](size_t i) {
              const float a = x[i] - k;
              s0 += a;
              s1 = fmaf(a, a, s1);
            });
  float sa, sb;
  block_sum2(s0 + t0, s1 + t1, sa, sb);
  const float d = sa / count;
  const float m = k + d;
  const float var = fmaxf(sb / count - d * d, 0.f);
  const float is = rsqrtf(var + eps);
  const float g = gamma[c], bt = beta[c];
  if (threadIdx.x == 0) {
    save_mean[c] = m;
    save_invstd[c] = is;
    if (running_mean) {
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * m;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * var * (count / (count - 1.f));
    }
    if (c == 0 && num_batches_tracked) *num_batches_tracked += 1;
    if (table) { table[4 * c + 0] = m; table[4 * c + 1] = is; table[4 * c + 2] = g; table[4 * c + 3] = bt; }
  }
  if (!y) return;
  const float lo = relu ? 0.f : -INFINITY;
  chan_slab(C, HW, c, 0, B,
            [&](size_t i) {
              const float4 v = *reinterpret_cast<const float4*>(x + i);
              float4 o;
              o.x = fmaxf(bn_affine(v.x, m, is, g, bt), lo); o.y = fmaxf(bn_affine(v.y, m, is, g, bt), lo);
              o.z = fmaxf(bn_affine(v.z, m, is, g, bt), lo); o.w = fmaxf(bn_affine(v.w, m, is, g, bt), lo);
              *reinterpret_cast<float4*>(y + i) = o;
            },
            [&](size_t i) { y[i] = fmaxf(bn_affine(x[i], m, is, g, bt), lo); });
}

__global__ __launch_bounds__(256) void k_bn_bwd_small(const float* __restrict__ x, const float* __restrict__ gy, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float* __restrict__ mean,
                                                       const float* __restrict__ invstd, float count, float* __restrict__ ggamma,
                                                       float* __restrict__ gbeta, float* __restrict__ gx, float* __restrict__ gx_chansum,
                                                       int B, int C, int HW, int relu) {
  const int c = blockIdx.x;
  const float m = mean[c], is = invstd[c], g = gamma[c], bt = beta[c];
  float s0 = 0.f, s1 = 0.f, t0 = 0.f, t1 = 0.f;
  auto acc = [&](float xv, float gv, float& a0, float& a1) {
    const float xh = bn_xhat(xv, m, is);
    if (relu && !(__fmaf_rn(xh, g, bt) > 0.f)) gv = 0.f;
    a0 += gv;
    a1 = fmaf(gv, xh, a1);
  };
  chan_slab(C, HW, c, 0, B,
            [&](size_t i) {
              const float4 v = *reinterpret_cast<const float4*>(x + i), w = *reinterpret_cast<const float4*>(gy + i);
              acc(v.x, w.x, s0, s1); acc(v.y, w.y, t0, t1); acc(v.z, w.z, s0, s1); acc(v.w, w.w, t0, t1);
            },
            [&](size_t i) { acc(x[i], gy[i], s0, s1); });
  float sa, sb;
  block_sum2(s0 + t0, s1 + t1, sa, sb);
  if (threadIdx.x == 0) { gbeta[c] = sa; ggamma[c] = sb; }
  const float ic = 1.f / count, ca = sa * ic, cb = sb * ic, sc = g * is;
  auto one = [&](float xv, float gv) {
    const float xh = bn_xhat(xv, m, is);
    if (relu && !(__fmaf_rn(xh, g, bt) > 0.f)) gv = 0.f;
    return sc * (gv - ca - xh * cb);
  };
  float u0 = 0.f, u1 = 0.f;                          // channel sum of gx = the bias gradient of the layer that produced x
  chan_slab(C, HW, c, 0, B,
            [&](size_t i) {
              const float4 v = *reinterpret_cast<const float4*>(x + i), w = *reinterpret_cast<const float4*>(gy + i);
              float4 o;
              o.x = one(v.x, w.x); o.y = one(v.y, w.y); o.z = one(v.z, w.z); o.w = one(v.w, w.w);
              *reinterpret_cast<float4*>(gx + i) = o;
              u0 += o.x; u1 += o.y; u0 += o.z; u1 += o.w;
            },
            [&](size_t i) {
              const float o = one(x[i], gy[i]);
              gx[i] = o;
              u0 += o;
            });
  if (gx_chansum) {
    float ua, ub;
    block_sum2(u0 + u1, 0.f, ua, ub);
    if (threadIdx.x == 0) gx_chansum[c] = ua;
  }
}

// evaluation mode (module.eval(): the --pretrained path of main.py:157-163 freezes the VAE this way): running statistics,
// y = relu?((x - rm) rsqrt(rv + eps) gamma + beta); backward w.r.t. x only (gx = g gamma invstd under the same mask)
__global__ __launch_bounds__(256) void k_bn_eval(const float* __restrict__ x, const float* __restrict__ gy, const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, const float* __restrict__ rmean,
                                                  const float* __restrict__ rvar, float eps, float* __restrict__ out, int B, int C, int HW,
                                                  int bps, int relu) {
  const int c = blockIdx.x, b0 = blockIdx.y * bps, nb = min(B, b0 + bps) - b0;
  const float m = rmean[c], is = rsqrtf(rvar[c] + eps), g = gamma[c], bt = beta[c];
  const float lo = relu ? 0.f : -INFINITY, sc = g * is;
  auto one = [&](float xv, float gv) {
    const float v = bn_affine(xv, m, is, g, bt);
    if (!gy) return fmaxf(v, lo);
    return (relu && !(v > 0.f)) ? 0.f : sc * gv;
  };
  chan_slab(C, HW, c, b0, nb,
            [&](size_t i) {
              const float4 v = *reinterpret_cast<const float4*>(x + i);
              const float4 w = gy ? *reinterpret_cast<const float4*>(gy + i) : float4{0.f, 0.f, 0.f, 0.f};
              float4 o;
              o.x = one(v.x, w.x); o.y = one(v.y, w.y); o.z = one(v.z, w.z); o.w = one(v.w, w.w);
              *reinterpret_cast<float4*>(out + i) = o;
            },
            [&](size_t i) { out[i] = one(x[i], gy ? gy[i] : 0.f); });
}

// part[split][c] = {sum v, 0}
__global__ __launch_bounds__(256) void k_chan_sum(const float* __restrict__ v, int B, int C, int HW, int bps, float* __restrict__ part) {
  const int c = blockIdx.x, b0 = blockIdx.y * bps, nb = min(B, b0 + bps) - b0;
  float s0 = 0.f, t0 = 0.f;
  chan_slab(C, HW, c, b0, nb,
            [&](size_t i) {
              const float4 a = *reinterpret_cast<const float4*>(v + i);
              s0 += a.x; t0 += a.y; s0 += a.z; t0 += a.w;
            },
            [&](size_t i) { s0 += v[i]; });
  const float in2[2] = {s0 + t0, 0.f};
  block_sum_store<2>(in2, part + ((size_t)blockIdx.y * C + c) * 2);
}

// out[c] = sum_s part[s][c][0]; one wavefront per channel
__global__ __launch_bounds__(64) void k_reduce_chan(const float* __restrict__ part, int nsplit, int C, float* __restrict__ out) {
  float a, b;
  split_sums(part, nsplit, C, blockIdx.x, a, b);
  if (threadIdx.x == 0) out[blockIdx.x] = a;
}


// ---- cross-rank BatchNorm (data parallelism: the statistics of the GLOBAL minibatch, vae.py:55,58,113,116,119) -------------
// local moments of this rank's shard: mom[2c] = mean_c, mom[2c+1] = M2_c = sum (x - mean_c)^2, mom[2C] = element count.
// One wavefront per channel; blockIdx.x == C writes the count.
__global__ __launch_bounds__(64) void k_bn_moments(const float* __restrict__ part, const float* __restrict__ shift, int nsplit, int C,
                                                   float count, float* __restrict__ mom) {
  const int c = blockIdx.x;
  if (c == C) { if (threadIdx.x == 0) mom[2 * C] = count; return; }
  float s0, s1;
  split_sums(part, nsplit, C, c, s0, s1);
  if (threadIdx.x != 0) return;
  const float d = s0 / count;
  mom[2 * c] = shift[c] + d;
  mom[2 * c + 1] = fmaxf(s1 - s0 * d, 0.f);
}

// gathered[r][2C+1] for r < W (<= 64) ranks, combined in rank order (Chan et al.: n = sum n_r, mean = sum n_r mean_r / n,
// M2 = sum M2_r + n_r (mean_r - mean)^2) -- identical on every rank, no E[x^2] - E[x]^2 cancellation.  Then exactly what
// k_bn_table does with the batch statistics.
__global__ __launch_bounds__(64) void k_bn_finalize(const float* __restrict__ gathered, int W, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, int C, float eps, float momentum,
                                                    float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                    float* __restrict__ running_mean, float* __restrict__ running_var,
                                                    long long* __restrict__ num_batches_tracked, float* __restrict__ table) {
  const int c = blockIdx.x, lane = threadIdx.x;
  const size_t stride = 2 * (size_t)C + 1;
  const float nr = lane < W ? gathered[lane * stride + 2 * C] : 0.f;
  const float mr = lane < W ? gathered[lane * stride + 2 * c] : 0.f;
  const float qr = lane < W ? gathered[lane * stride + 2 * c + 1] : 0.f;
  const float in2[2] = {nr, nr * mr};
  float out2[2];
  wave_sum_multi<2>(in2, out2);
  const float count = out2[0], m = out2[1] / count;
  const float dm = mr - m;
  const float in1[1] = {fmaf(nr * dm, dm, qr)};
  float out1[1];
  wave_sum_multi<1>(in1, out1);
  if (lane != 0) return;
  const float var = fmaxf(out1[0] / count, 0.f);
  const float is = rsqrtf(var + eps);
  save_mean[c] = m;
  save_invstd[c] = is;
  if (running_mean) {
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * m;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * var * (count / (count - 1.f));
  }
  if (c == 0 && num_batches_tracked) *num_batches_tracked += 1;
  table[4 * c + 0] = m; table[4 * c + 1] = is; table[4 * c + 2] = gamma[c]; table[4 * c + 3] = beta[c];
}

// y = relu?(affine(x)) with the per-channel table {mean, invstd, gamma, beta}: the output pass of k_bn_apply on its own
__global__ __launch_bounds__(256) void k_bn_apply_table(const float* __restrict__ x, const float* __restrict__ table, float* __restrict__ y,
                                                         int B, int C, int HW, int bps, int relu) {
  const int c = blockIdx.x, b0 = blockIdx.y * bps, nb = min(B, b0 + bps) - b0;
  const float m = table[4 * c], is = table[4 * c + 1], g = table[4 * c + 2], bt = table[4 * c + 3];
  const float lo = relu ? 0.f : -INFINITY;
  chan_slab(C, HW, c, b0, nb,
            [&](size_t i) {
              const float4 v = *reinterpret_cast<const float4*>(x + i);
              float4 o;
              o.x = fmaxf(bn_affine(v.x, m, is, g, bt), lo); o.y = fmaxf(bn_affine(v.y, m, is, g, bt), lo);
              o.z = fmaxf(bn_affine(v.z, m, is, g, bt), lo); o.w = fmaxf(bn_affine(v.w, m, is, g, bt), lo);
              *reinterpret_cast<float4*>(y + i) = o;
            },
            [&](size_t i) { y[i] = fmaxf(bn_affine(x[i], m, is, g, bt), lo); });
}

// out[c][0..1] = sum_s part[s][c][0..1]; one wavefront per channel
__global__ __launch_bounds__(64) void k_reduce_chan2(const float* __restrict__ part, int nsplit, int C, float* __restrict__ out) {
  float a, b;
  split_sums(part, nsplit, C, blockIdx.x, a, b);
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = a; out[2 * blockIdx.x + 1] = b; }
}

struct Split { int ns, bps, used; };
inline Split pick(int B) {
  Split s;
  s.ns = B < 64 ? B : 64;
  s.bps = (B + s.ns - 1) / s.ns;
  s.used = (B + s.bps - 1) / s.bps;
  return s;
}
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
// elements per channel up to which ONE workgroup per channel does the whole layer in one launch (GPODE_BN_ONE_LAUNCH=0: never)
// (configs[0], us, one launch vs two: forward 18k elements per channel 6.0 vs 9.9, 6k 5.6 vs 9.1, 1.5k 5.3 vs 8.2, but 86k 41.8 vs 10;
//  backward 1.5k 5.1 vs 9.9, 6k 12.7 vs 10.3, 18k 11.5 vs 11 -- a lone workgroup streams at a few GB/s)
constexpr size_t kBnSmallFwd = 20000, kBnSmallBwd = 2048;
inline bool bn_small(size_t per_channel, size_t limit) {
  static const bool off = [] { const char* e = getenv("GPODE_BN_ONE_LAUNCH"); return e && e[0] == '0'; }();
  return !off && per_channel <= limit;
}

}  // namespace

// floats: part[64][C][2], (unused [C][2]), shift[C], part_gx[64][C][2]
size_t bn_scratch(int B, int C) { return (size_t)(B < 64 ? B : 64) * C * 4 + (size_t)C * 3; }

int bn_fwd(const float* x, const float* gamma, const float* beta, float* y, float* save_mean, float* save_invstd,
           float* running_mean, float* running_var, long long* num_batches_tracked, float momentum, float eps, int B, int C, int HW,
           int relu, float* scratch, hipStream_t st) {
  if ((HW & 3) == 0 && !(aligned16(x) && aligned16(y))) return set_error("gpode_bn_fwd: x / y must be 16-byte aligned");
  if (bn_small((size_t)B * HW, kBnSmallFwd)) {
    hipLaunchKernelGGL(k_bn_fwd_small, C, 256, 0, st, x, gamma, beta, (float)B * HW, eps, momentum, save_mean, save_invstd, running_mean, running_var,
                       num_batches_tracked, y, (float*)nullptr, B, C, HW, relu);
    return check_launch("bn_fwd (one launch)");
  }
  const Split sp = pick(B);
  float* shift = scratch + (size_t)sp.ns * C * 2 + (size_t)C * 2;
  hipLaunchKernelGGL(k_bn_stats, dim3(C, sp.used), 256, 0, st, x, B, C, HW, sp.bps, scratch, shift);
  hipLaunchKernelGGL(k_bn_apply, dim3(C, sp.used), 256, 0, st, x, gamma, beta, scratch, shift, sp.used, (float)B * HW, eps, momentum, save_mean,
                     save_invstd, running_mean, running_var, num_batches_tracked, y, B, C, HW, sp.bps, relu);
  return check_launch("bn_fwd");
}

int bn_stats(const float* x, const float* gamma, const float* beta, float* save_mean, float* save_invstd, float* running_mean,
             float* running_var, long long* num_batches_tracked, float momentum, float eps, float* table, int B, int C, int HW,
             float* scratch, hipStream_t st) {
  if ((HW & 3) == 0 && !aligned16(x)) return set_error("gpode_bn_stats: x must be 16-byte aligned");
  if (bn_small((size_t)B * HW, kBnSmallFwd)) {
    hipLaunchKernelGGL(k_bn_fwd_small, C, 256, 0, st, x, gamma, beta, (float)B * HW, eps, momentum, save_mean, save_invstd, running_mean, running_var,
                       num_batches_tracked, (float*)nullptr, table, B, C, HW, 0);
    return check_launch("bn_stats (one launch)");
  }
  const Split sp = pick(B);
  float* shift = scratch + (size_t)sp.ns * C * 2 + (size_t)C * 2;
  hipLaunchKernelGGL(k_bn_stats, dim3(C, sp.used), 256, 0, st, x, B, C, HW, sp.bps, scratch, shift);
  hipLaunchKernelGGL(k_bn_table, C, 64, 0, st, gamma, beta, scratch, shift, sp.used, C, (float)B * HW, eps, momentum, save_mean, save_invstd,
                     running_mean, running_var, num_batches_tracked, table);
  return check_launch("bn_stats");
}

// the forward output is not needed: the ReLU mask is recomputed from x (same pinned arithmetic as k_bn_apply)
int bn_bwd(const float* x, const float* gy, const float* gamma, const float* beta, const float* save_mean,
           const float* save_invstd, float* gx, float* ggamma, float* gbeta, float* gx_chansum, int B, int C, int HW, int relu,
           float* scratch, hipStream_t st) {
  if ((HW & 3) == 0 && !(aligned16(x) && aligned16(gy) && aligned16(gx))) return set_error("gpode_bn_bwd: x / gy / gx must be 16-byte aligned");
  if (bn_small((size_t)B * HW, kBnSmallBwd)) {
    hipLaunchKernelGGL(k_bn_bwd_small, C, 256, 0, st, x, gy, gamma, beta, save_mean, save_invstd, (float)B * HW, ggamma, gbeta, gx, gx_chansum, B, C,
                       HW, relu);
    return check_launch("bn_bwd (one launch)");
  }
  const Split sp = pick(B);
  hipLaunchKernelGGL(k_bn_bwd_sums, dim3(C, sp.used), 256, 0, st, x, gy, gamma, beta, save_mean, save_invstd, B, C, HW, sp.bps, relu, scratch);
  float* part_gx = gx_chansum ? scratch + (size_t)sp.ns * C * 2 + (size_t)C * 3 : nullptr;
  hipLaunchKernelGGL(k_bn_bwd_apply, dim3(C, sp.used), 256, 0, st, x, gy, gamma, beta, save_mean, save_invstd, scratch, sp.used, (float)B * HW,
                     ggamma, gbeta, gx, part_gx, B, C, HW, sp.bps, relu, (const float*)nullptr, (const float*)nullptr, 0);
  if (gx_chansum && reduce_job(RedJob{part_gx, gx_chansum, sp.used, C, 2, 0, 0, 0}, st)) return 1;
  return check_launch("bn_bwd");
}

// ---- the same layer split at the points where ranks exchange statistics (include/gpode.h: "BatchNorm across ranks") ----------
int bn_moments(const float* x, float* mom, int B, int C, int HW, float* scratch, hipStream_t st) {
  if ((HW & 3) == 0 && !aligned16(x)) return set_error("gpode_bn_moments: x must be 16-byte aligned");
  const Split sp = pick(B);
  float* shift = scratch + (size_t)sp.ns * C * 2 + (size_t)C * 2;
  hipLaunchKernelGGL(k_bn_stats, dim3(C, sp.used), 256, 0, st, x, B, C, HW, sp.bps, scratch, shift);
  hipLaunchKernelGGL(k_bn_moments, C + 1, 64, 0, st, scratch, shift, sp.used, C, (float)B * HW, mom);
  return check_launch("bn_moments");
}

int bn_finalize(const float* gathered, int W, const float* gamma, const float* beta, float* save_mean, float* save_invstd,
                float* running_mean, float* running_var, long long* num_batches_tracked, float momentum, float eps, float* table, int C,
                hipStream_t st) {
  if (W < 1 || W > 64) return set_error("gpode_bn_finalize: %d ranks (1..64 supported)", W);
  hipLaunchKernelGGL(k_bn_finalize, C, 64, 0, st, gathered, W, gamma, beta, C, eps, momentum, save_mean, save_invstd, running_mean, running_var,
                     num_batches_tracked, table);
  return check_launch("bn_finalize");
}

int bn_apply(const float* x, const float* table, float* y, int B, int C, int HW, int relu, hipStream_t st) {
  if ((HW & 3) == 0 && !(aligned16(x) && aligned16(y))) return set_error("gpode_bn_apply: x / y must be 16-byte aligned");
  const Split sp = pick(B);
  hipLaunchKernelGGL(k_bn_apply_table, dim3(C, sp.used), 256, 0, st, x, table, y, B, C, HW, sp.bps, relu);
  return check_launch("bn_apply");
}

int bn_bwd_sums(const float* x, const float* gy, const float* gamma, const float* beta, const float* save_mean, const float* save_invstd,
                float* sums, int B, int C, int HW, int relu, float* scratch, hipStream_t st) {
  if ((HW & 3) == 0 && !(aligned16(x) && aligned16(gy))) return set_error("gpode_bn_bwd_sums: x / gy must be 16-byte aligned");
  const Split sp = pick(B);
  hipLaunchKernelGGL(k_bn_bwd_sums, dim3(C, sp.used), 256, 0, st, x, gy, gamma, beta, save_mean, save_invstd, B, C, HW, sp.bps, relu, scratch);
  hipLaunchKernelGGL(k_reduce_chan2, C, 64, 0, st, scratch, sp.used, C, sums);
  return check_launch("bn_bwd_sums");
}

// after gpode_bn_bwd_sums on the same scratch (its split sums are still there): gathered = every rank's sums, wts their weights
// relative to this rank, count = global element count
int bn_bwd_apply(const float* x, const float* gy, const float* gamma, const float* beta, const float* save_mean, const float* save_invstd,
                 const float* gathered, const float* wts, int W, float count, float* gx, float* ggamma, float* gbeta, float* gx_chansum, int B, int C, int HW, int relu,
                 float* scratch, hipStream_t st) {
  if ((HW & 3) == 0 && !(aligned16(x) && aligned16(gy) && aligned16(gx))) return set_error("gpode_bn_bwd_apply: x / gy / gx must be 16-byte aligned");
  if (W < 1 || W > 64) return set_error("gpode_bn_bwd_apply: %d ranks (1..64 supported)", W);
  const Split sp = pick(B);
  float* part_gx = gx_chansum ? scratch + (size_t)sp.ns * C * 2 + (size_t)C * 3 : nullptr;
  hipLaunchKernelGGL(k_bn_bwd_apply, dim3(C, sp.used), 256, 0, st, x, gy, gamma, beta, save_mean, save_invstd, scratch, sp.used, count,
                     ggamma, gbeta, gx, part_gx, B, C, HW, sp.bps, relu, gathered, wts, W);
  if (gx_chansum && reduce_job(RedJob{part_gx, gx_chansum, sp.used, C, 2, 0, 0, 0}, st)) return 1;
  return check_launch("bn_bwd_apply");
}

// gy == nullptr: forward (out = y); otherwise out = gx
int bn_eval(const float* x, const float* gy, const float* gamma, const float* beta, const float* running_mean, const float* running_var,
            float eps, float* out, int B, int C, int HW, int relu, hipStream_t st) {
  if ((HW & 3) == 0 && !(aligned16(x) && aligned16(out) && (!gy || aligned16(gy)))) return set_error("gpode_bn_eval: tensors must be 16-byte aligned");
  const Split sp = pick(B);
  hipLaunchKernelGGL(k_bn_eval, dim3(C, sp.used), 256, 0, st, x, gy, gamma, beta, running_mean, running_var, eps, out, B, C, HW, sp.bps, relu);
  return check_launch("bn_eval");
}

// out[c] = sum over (b, hw) of v[b,c,hw]   (bias gradient of ConvTranspose2d / Conv2d)
// ---------------------------------------------------------------------------------------------
// Final reductions of per-workgroup partial sums, many tensors in ONE launch.
// Every split-K weight gradient, bias gradient and BatchNorm channel sum of the backward pass ends in "sum the partials of nsplit
// workgroups in a fixed order" -- a launch of a few microseconds of work that costs a graph node (~5 us on the queue) each, 10-15
// per training step.  Between gpode_defer_reductions(1) and (0) those launches are recorded instead (their partials must stay alive),
// and gpode_flush_reductions() runs them together: grid (blocks, jobs), 64 outputs x 16 split groups per workgroup, LDS combine.
//   kind 0  out[e] = sum_s part[s n + e]
//   kind 1  the accumulator layout of k_convT_wgrad_mfma: part[s][tap][mt][nt][r][lane] -> gw[ci][co][tap]   (a = MT, b = NT, c = KK)
//   kind 2  out[e] = sum_s part[(s n + e) 2]      (channel sums kept as {sum, unused} pairs)
// ---------------------------------------------------------------------------------------------
namespace {
constexpr int kMaxRedJobs = 24;
struct RedJobs { RedJob j[kMaxRedJobs]; };
std::mutex g_red_mu;
RedJob g_red_jobs[kMaxRedJobs];
int g_red_n = 0, g_red_defer = 0;

__global__ __launch_bounds__(1024) void k_reduce_multi(RedJobs jobs) {
  __shared__ float red[16][64];
  const RedJob& jb = jobs.j[blockIdx.y];
  const int n = jb.n;
  if ((int)blockIdx.x * 64 >= n) return;
  const int ex = threadIdx.x & 63, sg = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + ex;
  float acc = 0.f;
  if (e < n) {
    const float* p = jb.part;
    if (jb.kind == 2) {
#pragma unroll 4
      for (int s = sg; s < jb.nsplit; s += 16) acc += p[((size_t)s * n + e) * 2];
    } else {
#pragma unroll 8
      for (int s = sg; s < jb.nsplit; s += 16) acc += p[(size_t)s * n + e];
    }
  }
  red[sg][ex] = acc;
  __syncthreads();
  if (sg == 0 && e < n) {
    float v = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) v += red[g][ex];
    if (jb.kind == 1) {
      const int MT = jb.a, NT = jb.b, KK = jb.c;
      const int lane = e & 63, r = (e >> 6) & 3, tile = e >> 8;       // tile = (tap * MT + mt) * NT + nt
      const int nt = tile % NT, mt = (tile / NT) % MT, t = tile / (NT * MT);
      const int ci = mt * 16 + 4 * (lane >> 4) + r, co = nt * 16 + (lane & 15);
      jb.out[((size_t)ci * (NT * 16) + co) * KK + t] = v;
    } else {
      jb.out[e] = v;
    }
  }
}

int launch_red(const RedJob* jobs, int nj, hipStream_t st) {
  RedJobs all;
  int maxn = 0;
  for (int i = 0; i < nj; ++i) { all.j[i] = jobs[i]; maxn = jobs[i].n > maxn ? jobs[i].n : maxn; }
  hipLaunchKernelGGL(k_reduce_multi, dim3((maxn + 63) / 64, nj), 1024, 0, st, all);
  return check_launch("reduce_multi");
}
}  // namespace

// several jobs of one producer in one launch, never deferred apart
int reduce_jobs(const RedJob* jobs, int nj, hipStream_t st) {
  if (nj < 1 || nj > kMaxRedJobs) return set_error("reduce_jobs: 1 .. %d jobs", kMaxRedJobs);
  {
    std::lock_guard<std::mutex> lk(g_red_mu);
    if (g_red_defer > 0 && g_red_n + nj <= kMaxRedJobs) {
      for (int i = 0; i < nj; ++i) g_red_jobs[g_red_n++] = jobs[i];
      return 0;
    }
  }
  return launch_red(jobs, nj, st);
}

int reduce_job(const RedJob& job, hipStream_t st) {
  if (job.n <= 0 || job.nsplit <= 0) return set_error("reduce_job: empty job");
  {
    std::lock_guard<std::mutex> lk(g_red_mu);
    if (g_red_defer > 0 && g_red_n < kMaxRedJobs) {
      g_red_jobs[g_red_n++] = job;
      return 0;
    }
  }
  return launch_red(&job, 1, st);
}
// mode 1: record from now on (nests); 0: one level back; 2: drop whatever an aborted backward pass left behind, then as 1
void defer_reductions(int mode) {
  std::lock_guard<std::mutex> lk(g_red_mu);
  if (mode == 2) { g_red_n = 0; g_red_defer = 0; }
  if (mode) ++g_red_defer;
  else if (g_red_defer > 0) --g_red_defer;
}
int flush_reductions(hipStream_t st) {
  RedJob local[kMaxRedJobs];
  int nj;
  {
    std::lock_guard<std::mutex> lk(g_red_mu);
    nj = g_red_n;
    for (int i = 0; i < nj; ++i) local[i] = g_red_jobs[i];
    g_red_n = 0;
  }
  return nj ? launch_red(local, nj, st) : 0;
}

int chan_sum(const float* v, float* out, int B, int C, int HW, float* scratch, hipStream_t st) {
  if ((HW & 3) == 0 && !aligned16(v)) return set_error("gpode_chan_sum: v must be 16-byte aligned");
  const Split sp = pick(B);
  hipLaunchKernelGGL(k_chan_sum, dim3(C, sp.used), 256, 0, st, v, B, C, HW, sp.bps, scratch);
  if (reduce_job(RedJob{scratch, out, sp.used, C, 2, 0, 0, 0}, st)) return 1;
  return check_launch("chan_sum");
}

}  // namespace gp
