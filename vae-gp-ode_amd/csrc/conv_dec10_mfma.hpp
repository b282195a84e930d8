// conv_dec10_mfma.hpp -- the decoder's last layer, ConvTranspose2d(16 -> 1, K = 5, stride 1, padding 2) on 28 x 28 images
// (reference: experiments/model/core/vae.py:80-84, decnn.10), on the fp32 matrix cores.  With one output channel the
// channel dimension cannot fill an MFMA tile, so the 25 taps take that role:
//
//   forward   T[tap][p'] = sum_ci w[ci][tap] x[ci][p']           (MFMA: m = tap, n = source pixel, k = ci)
//             y[oy][ox]  = b + sum_tap T[tap][oy + 2 - ky][ox + 2 - kx]   (shift-and-add through LDS)
//   d/d x     gx[ci][p]  = sum_tap w[ci][tap] gy[window(p, tap)]  (MFMA: m = ci, n = pixel, k = tap, 25 -> 28)
//   d/d w     gw[ci][tap] = sum_{b,p} x[b][ci][p] gy[b][window(p, tap)]   (MFMA: m = ci, n = tap, k = pixel)
//
// x is read from global memory straight into MFMA operands (every element is used by exactly one k-slot);
// only the single-channel gy / T planes go through LDS, stored as zero-bordered 32 x 32 planes (index = o + 2).
// Persistent grids (<= one 512-thread workgroup per CU, two wavefronts per SIMD).
#pragma once
#include "conv_mfma.hpp"
#include "bn_math.hpp"
#include "wave_reduce.hpp"

namespace gp {
namespace dec10 {

constexpr int CI = 16, H = 28, NP = H * H, KK = 25, WP = 32, PLANE = WP * WP;
constexpr int NTILE = NP / 16;                       // 49 tiles of 16 pixels
constexpr int PST = PLANE + 4;                       // T plane stride: 4 planes apart == 16 banks apart

// ---------------------------------------------------------------------------------------------
// forward.  grid <= 256, block 512, LDS = 25 * PST floats.
// ---------------------------------------------------------------------------------------------
template <bool HAS_BN>
__global__ __launch_bounds__(512) void k_fwd(const float* __restrict__ x, const float* __restrict__ w,
                                             const float* __restrict__ bias, float* __restrict__ y, int B,
                                             const float* __restrict__ in_bn) {
  float* s_T = igemm_smem;                           // [25][PST], zero borders
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lk = lane >> 4;
  constexpr int MAXT = (NTILE + 7) / 8;              // tiles per wavefront: t = wave + 8 j
  for (int e = tid; e < KK * PST; e += 512) s_T[e] = 0.f;
  // A = w[ci = 4 ks + lk][tap = 16 c + lr]
  float wa[4][2];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int tap = 16 * c + lr;
      wa[ks][c] = tap < KK ? w[(size_t)(4 * ks + lk) * KK + tap] : 0.f;
    }
  const float bv = bias ? bias[0] : 0.f;
  // in_bn ([16][4] = mean, invstd, gamma, beta): x is the raw output of decnn.7 and decnn.8/9 (BatchNorm + ReLU) are applied
  // to the operands in registers
  float4 tf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) tf[ks] = HAS_BN ? reinterpret_cast<const float4*>(in_bn)[4 * ks + lk] : float4{0.f, 1.f, 1.f, 0.f};
  // B = x[b][ci = 4 ks + lk][16 t + lr], fetched one image ahead
  float xb[MAXT][4];
  auto prefetch = [&](int b) {
    const float* xp = x + (size_t)b * CI * NP + lr;
#pragma unroll
    for (int j = 0; j < MAXT; ++j) {
      const int t = wave + 8 * j;
      if (t < NTILE) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) xb[j][ks] = xp[(size_t)(4 * ks + lk) * NP + 16 * t];
      }
    }
  };
  if ((int)blockIdx.x < B) prefetch(blockIdx.x);
  __syncthreads();
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
#pragma unroll
    for (int j = 0; j < MAXT; ++j) {
      const int t = wave + 8 * j;
      if (t < NTILE) {                               // wave-uniform
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const float xv = HAS_BN ? bn_relu(xb[j][ks], tf[ks]) : xb[j][ks];
#pragma unroll
          for (int c = 0; c < 2; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[ks][c], xv, acc[c], 0, 0, 0);
        }
        // lane: source pixel 16 t + lr, taps 16 c + 4 lk + r
        const int p = 16 * t + lr, pa = (p / H + 2) * WP + p % H + 2;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int tap = 16 * c + 4 * lk + r;
            if (tap < KK) s_T[tap * PST + pa] = acc[c][r];
          }
      }
    }
    if (b + (int)gridDim.x < B) prefetch(b + gridDim.x);
    __syncthreads();
    for (int o = tid; o < NP; o += 512) {
      const int oy = o / H, ox = o % H;
      const float* tp = s_T + (oy + 4) * WP + ox + 4;                // stored row oy + 4 - ky, col ox + 4 - kx
      float s0 = bv, s1 = 0.f;
#pragma unroll
      for (int ky = 0; ky < 5; ++ky)
#pragma unroll
        for (int kx = 0; kx < 5; ++kx) {
          const float v = tp[(ky * 5 + kx) * PST - ky * WP - kx];
          if ((ky * 5 + kx) & 1) s1 += v; else s0 += v;
        }
      y[(size_t)b * NP + o] = s0 + s1;
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// d/d input.  grid <= 256, block 512, LDS = IPB planes.
// ---------------------------------------------------------------------------------------------
template <int IPB>
__global__ __launch_bounds__(512) void k_bwd_data(const float* __restrict__ gy, const float* __restrict__ w, float* __restrict__ gx, int B) {
  float* s_g = igemm_smem;                           // [IPB][32][32], index = o + 2, zero borders
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lk = lane >> 4;
  constexpr int NKS = (KK + 3) / 4;                  // 7 k-steps of 4 taps
  constexpr int NLD = (IPB * NP / 4 + 511) / 512;
  for (int e = tid; e < IPB * PLANE; e += 512) s_g[e] = 0.f;
  // A = w[ci = lr][tap = 4 ks + lk]; B gathers the window element of the lane's tap
  float wa[NKS];
  int toff[NKS];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    const int tap = 4 * ks + lk;
    wa[ks] = tap < KK ? w[(size_t)lr * KK + tap] : 0.f;
    toff[ks] = tap < KK ? (tap / 5) * WP + tap % 5 : 0;
  }
  const int ngroups = (B + IPB - 1) / IPB;
  float4 pre[NLD];
  auto prefetch = [&](int grp) {
    const int b0 = grp * IPB, nf4 = min(IPB, B - b0) * (NP / 4);
    const float4* src = reinterpret_cast<const float4*>(gy) + (size_t)b0 * (NP / 4);
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int f = tid + 512 * i;
      if (f < nf4) pre[i] = src[f];
    }
  };
  if ((int)blockIdx.x < ngroups) prefetch(blockIdx.x);
  for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int b0 = grp * IPB, nimg = min(IPB, B - b0);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int f = tid + 512 * i;
      if (f < nimg * (NP / 4)) {
        const float v[4] = {pre[i].x, pre[i].y, pre[i].z, pre[i].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int e = 4 * f + k, im = e / NP, q = e % NP;
          s_g[im * PLANE + (q / H + 2) * WP + q % H + 2] = v[k];
        }
      }
    }
    __syncthreads();
    if (grp + (int)gridDim.x < ngroups) prefetch(grp + gridDim.x);
    const int ntiles = nimg * NTILE;
    for (int t = wave; t < ntiles; t += 8) {
      const int im = t / NTILE, p = (t % NTILE) * 16 + lr;
      const float* gp = s_g + im * PLANE + (p / H) * WP + p % H;     // window origin (iy + ky, ix + kx)
      float bf[NKS];
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) bf[ks] = gp[toff[ks]];
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[ks], bf[ks], acc, 0, 0, 0);
      // lane: pixel p, channels 4 lk + r
      float* op = gx + ((size_t)(b0 + im) * CI + 4 * lk) * NP + p;
#pragma unroll
      for (int r = 0; r < 4; ++r) op[(size_t)r * NP] = acc[r];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// d/d input FUSED with the backward of the BatchNorm + ReLU in front of this layer (vae.py:119-120 feeding :121).  The
// gradient w.r.t. the normalised activation, ga = d/d input above, costs 25 multiply-adds per element from a 3 KB plane of gy,
// while writing it and reading it back twice (once for the BatchNorm sums, once for the BatchNorm input gradient) moves
// 3 x 16 x 784 floats per image through HBM -- so it is recomputed in both passes and never stored:
//   MODE 0  part[wg][c] = {sum g, sum g xhat},  g = ga masked by the ReLU        (reads c = the BatchNorm input, gy)
//   MODE 1  gc = gamma invstd (g - mean(g) - xhat mean(g xhat)),  part_gx[wg][c] = sum gc   (reads c, gy; writes gc)
// Same tile loop as k_bwd_data; c is fetched NT tiles ahead (4 NT loads in flight per lane).  2 workgroups per CU.
// ---------------------------------------------------------------------------------------------
constexpr int BN_NT = 4;                           // (8 measured: no change -- the passes are not latency-bound)

// sA[c], sB[c] <- sums over the workgroups' partials part[wg][c][0..1], identical in every workgroup (fixed order)
__device__ __forceinline__ void reduce_parts(const float* __restrict__ part, int nsplit, float* sA, float* sB) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int c = 2 * wave + q;
    float a = 0.f, b = 0.f;
    for (int sp = lane; sp < nsplit; sp += 64) {
      a += part[((size_t)sp * CI + c) * 2];
      b += part[((size_t)sp * CI + c) * 2 + 1];
    }
    const float in2[2] = {a, b};
    float out2[2];
    wave_sum_multi<2>(in2, out2);
    if (lane == 0) { sA[c] = out2[0]; sB[c] = out2[1]; }
  }
}

// WGRAD (MODE 0 only): the layer's WEIGHT gradient rides in the sums pass -- it reads the same c (as relu(bn(c)), the layer's input) and
// the same gy plane, in the same lane layout k_wgrad uses (A = the lane's four pixels of channel lr, B = the gy window of tap 16 c + lr):
// 8 more MFMAs per tile in a pass that waits for HBM, instead of a pass of its own over c.  part_w[workgroup][CI * KK].
template <int IPB, int MODE, bool WGRAD = false>
__global__ __launch_bounds__(512, 4) void k_bwd_data_bn(const float* __restrict__ gy, const float* __restrict__ w, const float* __restrict__ x,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const float* __restrict__ mean, const float* __restrict__ invstd, int B,
                                                        float* __restrict__ part,                       // MODE 0: out; MODE 1: in
                                                        int nsplit, float count, const float* __restrict__ gathered,
                                                        const float* __restrict__ wts, int W, float* __restrict__ ggamma,
                                                        float* __restrict__ gbeta, float* __restrict__ gx, float* __restrict__ part_gx,
                                                        float* __restrict__ part_w = nullptr, float* __restrict__ part_b = nullptr) {
  static_assert(!WGRAD || MODE == 0, "the weight gradient rides in the sums pass");
  float* s_g = igemm_smem;                           // [IPB][32][32], index = o + 2, zero borders
  float* s_red = s_g + IPB * PLANE;                  // [8][4][CI][2]
  float* s_A = s_red + 8 * 4 * CI * 2;               // [CI] sum g  (MODE 1)
  float* s_B = s_A + CI;                             // [CI] sum g xhat
  float* s_wred = s_B + CI;                          // WGRAD: [8][CI][32] (the launch adds the room)
  int toffw[2];                                      // WGRAD: window offset of tap 16 c + lr (taps >= 25 alias tap 24, dropped at the end)
  f32x4 accw[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lk = lane >> 4;
  constexpr int NKS = (KK + 3) / 4;
  constexpr int NLD = (IPB * NP / 4 + 511) / 512;
  for (int e = tid; e < IPB * PLANE; e += 512) s_g[e] = 0.f;
  // The tile product is taken transposed with respect to k_bwd_data -- D[m = pixel][n = ci] = sum_tap G[pixel][tap] w[ci][tap] --
  // with the SAME operand registers swapped (A[m = lr][k = lk] and B[k = lk][n = lr] index a lane alike): a lane then holds 4
  // consecutive pixels of ONE channel, so c and gc move as 16-byte accesses and the channel's constants are 7 registers, not 28.
  float wb[NKS];                                     // B[k = tap][n = ci = lr]
  int toff[NKS];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    const int tap = 4 * ks + lk;
    wb[ks] = tap < KK ? w[(size_t)lr * KK + tap] : 0.f;
    toff[ks] = tap < KK ? (tap / 5) * WP + tap % 5 : 0;
  }
  const float cm = mean[lr], cis = invstd[lr], cg = gamma[lr], cbt = beta[lr], sc = cg * cis;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int tap = min(16 * c + lr, KK - 1);
    toffw[c] = (tap / 5) * WP + tap % 5;
  }
  float ca = 0.f, cb = 0.f;
  if (MODE == 1) {
    reduce_parts(part, nsplit, s_A, s_B);
    __syncthreads();
    if (blockIdx.x == 0 && tid < CI) { gbeta[tid] = s_A[tid]; ggamma[tid] = s_B[tid]; }   // this rank's own sums: the affine gradients
    if (gathered) {                                  // centring terms from the weighted sums over all ranks (gathered[r][2 CI])
      __syncthreads();
      if (tid < 2 * CI) {
        float v = 0.f;
        for (int rk = 0; rk < W; ++rk) v = fmaf(wts[rk], gathered[(size_t)rk * 2 * CI + tid], v);
        (tid & 1 ? s_B : s_A)[tid >> 1] = v;
      }
      __syncthreads();
    }
    const float ic = 1.f / count;
    ca = s_A[lr] * ic;
    cb = s_B[lr] * ic;
  }
  float a0 = 0.f, a1 = 0.f;
  float gsum = 0.f;                                  // WGRAD: this thread's share of sum gy = the layer's bias gradient (one output channel)
  const int ngroups = (B + IPB - 1) / IPB;
  float4 pre[NLD];
  auto prefetch = [&](int grp) {
    const int b0 = grp * IPB, nf4 = min(IPB, B - b0) * (NP / 4);
    const float4* src = reinterpret_cast<const float4*>(gy) + (size_t)b0 * (NP / 4);
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int f = tid + 512 * i;
      if (f < nf4) pre[i] = src[f];
    }
  };
  if ((int)blockIdx.x < ngroups) prefetch(blockIdx.x);
  for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int b0 = grp * IPB, nimg = min(IPB, B - b0);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int f = tid + 512 * i;
      if (f < nimg * (NP / 4)) {
        const float v[4] = {pre[i].x, pre[i].y, pre[i].z, pre[i].w};
        if (WGRAD) gsum += (v[0] + v[1]) + (v[2] + v[3]);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int e = 4 * f + k, im = e / NP, q = e % NP;
          s_g[im * PLANE + (q / H + 2) * WP + q % H + 2] = v[k];
        }
      }
    }
    __syncthreads();
    if (grp + (int)gridDim.x < ngroups) prefetch(grp + gridDim.x);
    const int ntiles = nimg * NTILE;
    for (int t0 = wave; t0 < ntiles; t0 += 8 * BN_NT) {
      float4 xv[BN_NT];
#pragma unroll
      for (int u = 0; u < BN_NT; ++u) {
        const int t = t0 + 8 * u;
        if (t < ntiles) {                            // wave-uniform
          const int im = t / NTILE, p0 = (t % NTILE) * 16 + 4 * lk;
          xv[u] = *reinterpret_cast<const float4*>(x + ((size_t)(b0 + im) * CI + lr) * NP + p0);
        }
      }
#pragma unroll
      for (int u = 0; u < BN_NT; ++u) {
        const int t = t0 + 8 * u;
        if (t < ntiles) {
          const int im = t / NTILE, pa = (t % NTILE) * 16 + lr;       // A: the window of pixel lr of the tile
          const float* gp = s_g + im * PLANE + (pa / H) * WP + pa % H;
          f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < NKS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(gp[toff[ks]], wb[ks], acc, 0, 0, 0);
          // D[m = pixel 4 lk + r of the tile][n = ci = lr]
          const float xs[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w};
          float o[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float xh = bn_xhat(xs[r], cm, cis);
            const float g = (__fmaf_rn(xh, cg, cbt) > 0.f) ? acc[r] : 0.f;
            if (MODE == 0) {
              a0 += g;
              a1 = fmaf(g, xh, a1);
            } else {
              o[r] = sc * (g - ca - xh * cb);
              a0 += o[r];
            }
          }
          if (MODE == 1)
            *reinterpret_cast<float4*>(gx + ((size_t)(b0 + im) * CI + lr) * NP + (t % NTILE) * 16 + 4 * lk) = float4{o[0], o[1], o[2], o[3]};
          if constexpr (WGRAD) {
            // gw[ci = lr][tap] += relu(bn(c))[ci][pixel] gy[window(pixel, tap)]: MFMA i takes the lane's i-th pixel as its k-slot
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int p = (t % NTILE) * 16 + 4 * lk + i;
              const float* gq = s_g + im * PLANE + (p / H) * WP + p % H;
              const float av = fmaxf(__fmaf_rn(bn_xhat(xs[i], cm, cis), cg, cbt), 0.f);
              accw[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, gq[toffw[0]], accw[0], 0, 0, 0);
              accw[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, gq[toffw[1]], accw[1], 0, 0, 0);
            }
          }
        }
      }
    }
  }
  // channel lr of this lane: the four row groups and the 8 wavefronts in a fixed order
  s_red[((wave * 4 + lk) * CI + lr) * 2] = a0;
  s_red[((wave * 4 + lk) * CI + lr) * 2 + 1] = a1;
  __syncthreads();
  float* outp = MODE == 0 ? part : part_gx;
  if (tid < 2 * CI && outp) {
    float v = 0.f;
#pragma unroll
    for (int q = 0; q < 32; ++q) v += s_red[q * CI * 2 + tid];
    outp[(size_t)blockIdx.x * CI * 2 + tid] = v;
  }
  if constexpr (WGRAD) {
    // D[m = ci][n = tap]: lane holds tap 16 c + lr, ci = 4 lk + r; the 8 wavefronts in a fixed order (as k_wgrad)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) s_wred[(wave * CI + 4 * lk + r) * 32 + 16 * c + lr] = accw[c][r];
    __syncthreads();
    for (int e = tid; e < CI * KK; e += 512) {
      const int ci = e / KK, tap = e % KK;
      float v = 0.f;
#pragma unroll
      for (int wv = 0; wv < 8; ++wv) v += s_wred[(wv * CI + ci) * 32 + tap];
      part_w[(size_t)blockIdx.x * (CI * KK) + e] = v;
    }
    if (part_b) {                                    // bias gradient: the 512 threads' shares in a fixed order
      __syncthreads();
      s_wred[tid] = gsum;
      __syncthreads();
      if (tid < 64) {
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) v += s_wred[tid + 64 * q];
        const float in1[1] = {v};
        float out1[1];
        wave_sum_multi<1>(in1, out1);
        if (tid == 0) part_b[blockIdx.x] = out1[0];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// d/d weight.  grid <= 256, block 512, LDS = max(IPB planes, reduction buffer); part[blockIdx.x][ci][tap].
// ---------------------------------------------------------------------------------------------
template <int IPB>
__global__ __launch_bounds__(512) void k_wgrad(const float* __restrict__ x, const float* __restrict__ gy, float* __restrict__ part, int B,
                                               const float* __restrict__ in_bn) {
  float* s_g = igemm_smem;                           // [IPB][32][32]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lk = lane >> 4;
  constexpr int NLD = (IPB * NP / 4 + 511) / 512;
  constexpr int NJ = (IPB * NTILE + 7) / 8;          // 16-pixel groups per wavefront per image group
  for (int e = tid; e < IPB * PLANE; e += 512) s_g[e] = 0.f;
  // B = gy window element of tap 16 c + lr (taps >= 25 alias tap 24; their columns are dropped at the end)
  int toff[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int tap = min(16 * c + lr, KK - 1);
    toff[c] = (tap / 5) * WP + tap % 5;
  }
  f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  const float4 tf = in_bn ? reinterpret_cast<const float4*>(in_bn)[lr] : float4{0.f, 1.f, 1.f, 0.f};   // BatchNorm + ReLU of x (channel lr)
  const int ngroups = (B + IPB - 1) / IPB;
  float4 pre[NLD];
  float4 xa[NJ];                                     // A = x[b][ci = lr][16 j + 4 lk + i], i = 0..3 -> MFMA i
  auto prefetch = [&](int grp) {
    const int b0 = grp * IPB, nimg = min(IPB, B - b0);
    const float4* src = reinterpret_cast<const float4*>(gy) + (size_t)b0 * (NP / 4);
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int f = tid + 512 * i;
      if (f < nimg * (NP / 4)) pre[i] = src[f];
    }
#pragma unroll
    for (int q = 0; q < NJ; ++q) {
      const int g = wave + 8 * q;                    // group index over (image, 16-pixel group)
      if (g < nimg * NTILE) {
        const int im = g / NTILE, j = g % NTILE;
        xa[q] = *reinterpret_cast<const float4*>(x + ((size_t)(b0 + im) * CI + lr) * NP + 16 * j + 4 * lk);
      }
    }
  };
  if ((int)blockIdx.x < ngroups) prefetch(blockIdx.x);
  for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int nimg = min(IPB, B - grp * IPB);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int f = tid + 512 * i;
      if (f < nimg * (NP / 4)) {
        const float v[4] = {pre[i].x, pre[i].y, pre[i].z, pre[i].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int e = 4 * f + k, im = e / NP, q = e % NP;
          s_g[im * PLANE + (q / H + 2) * WP + q % H + 2] = v[k];
        }
      }
    }
    __syncthreads();
    float4 xc[NJ];
#pragma unroll
    for (int q = 0; q < NJ; ++q) xc[q] = xa[q];
    if (grp + (int)gridDim.x < ngroups) prefetch(grp + gridDim.x);
#pragma unroll
    for (int q = 0; q < NJ; ++q) {
      const int g = wave + 8 * q;
      if (g < nimg * NTILE) {                        // wave-uniform
        const int im = g / NTILE, j = g % NTILE;
        float av[4] = {xc[q].x, xc[q].y, xc[q].z, xc[q].w};
        if (in_bn) {
#pragma unroll
          for (int i = 0; i < 4; ++i) av[i] = bn_relu(av[i], tf);
        }
        float bf[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int p = 16 * j + 4 * lk + i;
          const float* gp = s_g + im * PLANE + (p / H) * WP + p % H;
          bf[i][0] = gp[toff[0]];
          bf[i][1] = gp[toff[1]];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int c = 0; c < 2; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bf[i][c], acc[c], 0, 0, 0);
      }
    }
  }
  // D[m = ci][n = tap]: lane holds tap 16 c + lr, ci = 4 lk + r; combine the 8 wavefronts in a fixed order
  __syncthreads();
  float* s_red = igemm_smem;                         // [8][16][32]
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) s_red[(wave * CI + 4 * lk + r) * 32 + 16 * c + lr] = acc[c][r];
  __syncthreads();
  for (int e = tid; e < CI * KK; e += 512) {
    const int ci = e / KK, tap = e % KK;
    float v = 0.f;
#pragma unroll
    for (int wv = 0; wv < 8; ++wv) v += s_red[(wv * CI + ci) * 32 + tap];
    part[(size_t)blockIdx.x * (CI * KK) + e] = v;
  }
}

}  // namespace dec10
}  // namespace gp
